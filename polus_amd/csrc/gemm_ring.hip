// bf16 MFMA GEMM, 256x128 tile, 4 waves, 3-stage LDS-DMA ring, TWO workgroups per CU (gfx950).
// C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous (forward Dense; dX through the
// transposed weight shadow).
//
// Why two workgroups per CU: with one un-staggered 8-wave workgroup per CU (round 1's 256 x 256 kernel, removed) the per-tile
// prologue (first loads), epilogue (bias/GELU/residual + C stores, which also have to drain
// before the next counted wait because vmcnt retires in order) and the LDS-read part of every
// phase leave the matrix pipe idle, and K = 768 gives only 12 K-steps per tile to amortise them.
// Two independent 4-wave workgroups (72 KiB LDS, <= 256 VGPRs each) interleave on the 4 SIMDs:
// one's epilogue / LDS reads / barrier overlap the other's MFMAs.
//
// Per workgroup: waves 2(M) x 2(N), 128x64 per wave (8x4 MFMA 16x16x32 tiles), K-step 32 (one
// MFMA k-step), LDS stage = A 256 rows x 64 B | B 128 rows x 64 B = 24 KiB, ring of 3.  Loads
// run two K-steps ahead: step t issues tile t+2 (6 x global_load_lds_dwordx4 per wave) into
// the stage read in step t-1; the only wait is `s_waitcnt vmcnt(6)` + one raw s_barrier per
// K-step.  LDS-DMA writes are lane-linear (16 rows x 64 B per wave-instruction), so the bank
// swizzle sits on the source address: LDS chunk pc of row r holds K-chunk pc ^ pi((r>>2)&3),
// pi = [0,2,3,1]; the fragment read applies the same XOR and every ds_read_b128 lane group hits
// 16 distinct 16-byte slots.
//
// A_KS / B_KS select K-strided operands ([K][rows] in memory): W in dX = dY W, both dY and X in
// dW = dY^T X.  Their LDS
// images stay k-major (A: 32 k-rows x 512 B, B: 32 k-rows x 256 B; a DMA piece = 2 resp. 4 whole
// k-rows, full 128-byte lines from HBM), fragments come out of ds_read_b64_tr_b16, and the
// 32-byte unit index inside a k-row is XORed with (k&3) | ((k>>3)&1)<<2 on the DMA source side
// and on the read side, so the 8 k-rows a half-wave reads fall on 8 distinct 32-byte bank
// windows.  Split-K over blockIdx.y writes f32 slabs reduced by an order-fixed second kernel.
#include <cstddef>
#include "gemm_common.h"

using namespace pgemm;

namespace {

constexpr int TM = 256, TN = 128, TK = 32, NTHR = 256;
constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64;
constexpr int STAGE = A_BYTES + B_BYTES;   // 24 KiB
constexpr int NSTAGE = 3;
constexpr int SMEM_BYTES = NSTAGE * STAGE; // 72 KiB

__device__ const uint4 g_zero_chunk_ring[1] = {{0u, 0u, 0u, 0u}};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ __forceinline__ int pi4(int q) { return (0x78 >> (2 * q)) & 3; }

#define POLUS_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define POLUS_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// One 256x128 output tile (workgroup `wg` of the `nwg` that cover the problem) of K-split `split`.
// MODE >= 0 (bf16 C, both operands K-contiguous): epilogue variant fixed at compile time
// (pgemm::epilogue_wave); MODE = -1: the run-time epilogue (f32 C, split-K slabs, K-strided operands).
template <typename TC, bool A_KS, bool B_KS, bool DROP, int MODE = -1>
__device__ __forceinline__ void ring_body(const GemmArgs& p, const int wg_in, const int nwg, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = (p.N + TN - 1) / TN;
    const int wg = xcd_remap(wg_in, nwg);
    const int tile_m = wg / tiles_n, tile_n = wg % tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int kbeg = split * p.k_per_split;
    const int K = min(p.K, kbeg + p.k_per_split);   // this split's k range is [kbeg, K)
    const int nk = (K - kbeg + TK - 1) / TK;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_chunk_ring);

    // ---- per-lane DMA sources: wave w loads A rows 64w..64w+63 (4 pieces) and B rows 32w..32w+31
    // (2 pieces); lane l of a piece covers row (l>>2), LDS chunk (l&3)
    const bf16_t* src[6];
    int kofs[6];       // K offset of this lane's chunk within a K-step (elements for KC, k-rows for KS)
    bool rowok[6];
    int ldsofs[6];     // wave-uniform LDS byte offset of the piece within a stage
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const bool isA = j < 4;
        const bf16_t* base = static_cast<const bf16_t*>(isA ? p.A : p.B);
        const long ld = isA ? p.lda : p.ldb;
        const bool ks = isA ? A_KS : B_KS;
        if (!ks) {
            const int row = isA ? (64 * wid + 16 * j + (lane >> 2)) : (32 * wid + 16 * (j - 4) + (lane >> 2));
            const int lc = (lane & 3) ^ pi4((row >> 2) & 3);
            const int gr = (isA ? m0 : n0) + row;
            rowok[j] = gr < (isA ? p.M : p.N);
            kofs[j] = lc * 8;
            src[j] = base + (long)gr * ld + lc * 8;
            ldsofs[j] = isA ? (64 * wid + 16 * j) * 64 : A_BYTES + (32 * wid + 16 * (j - 4)) * 64;
        } else {
            // A: piece = 2 k-rows x 512 B, wave w owns k-rows 8w..8w+7; B: piece = 4 k-rows x 256 B
            const int krow = isA ? (8 * wid + 2 * j + (lane >> 5)) : (8 * wid + 4 * (j - 4) + (lane >> 4));
            const int pc = isA ? (lane & 31) : (lane & 15);
            const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int lc = pc ^ (f << 1);
            const int gr = (isA ? m0 : n0) + lc * 8;
            rowok[j] = gr < (isA ? p.M : p.N);
            kofs[j] = krow;
            src[j] = base + (long)krow * ld + gr;
            ldsofs[j] = isA ? (8 * wid + 2 * j) * 512 : A_BYTES + (8 * wid + 4 * (j - 4)) * 256;
        }
    }
    const long a_kstep = A_KS ? (long)p.lda : 1, b_kstep = B_KS ? (long)p.ldb : 1;
    auto issue = [&](int stage, int k0) {
        unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bf16_t* sp = (rowok[j] && k0 + kofs[j] < K) ? src[j] + (long)k0 * (j < 4 ? a_kstep : b_kstep) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lds_void_t*)(st + ldsofs[j]), 16, 0, 0);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // column sums of A (dW = dY^T X: sum_t dY[t][m] = bias gradient): D[n][m] = sum_k 1 * A[m][k]
    const bool want_colsum = A_KS && p.colsum_a != nullptr && wn == 0 && n0 == 0;
    f32x4 csum[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) csum[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag<bf16_t> ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones.v[e] = (bf16_t)1.0f;

    // fragment address: row R = base16 + i -> (R>>2)&3 = (i>>2)&3; logical chunk g
    const int fc = (g ^ pi4((i >> 2) & 3)) * 16;
    const int a_off = (wm * 128 + i) * 64 + fc;             // K-contiguous image: + mt*1024
    const int b_off = A_BYTES + (wn * 64 + i) * 64 + fc;    // + nt*1024
    // K-strided image: lane (i,g) passes k-row 8g + (i>>2) (+4 for the second half), columns 4*(i&3)..+3
    const int krow = 8 * g + (i >> 2);
    const int fx = ((krow & 3) | (((krow >> 3) & 1) << 2)) << 5;    // same for krow + 4

    // prologue: two tiles in flight, first one landed
    issue(0, kbeg);
    if (nk > 1) { issue(1, kbeg + TK); POLUS_VMCNT(6); } else { POLUS_VMCNT(0); }
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        const unsigned char* st = smem + stage * STAGE;
        const bool dma = t + 2 < nk;
        if (dma) issue(stage == 0 ? 2 : stage - 1, kbeg + (t + 2) * TK);   // stage read in step t-1
        Frag<bf16_t> af[8], bfr[4];
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        if (!B_KS) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bfr[nt].v = *reinterpret_cast<const bf16x8*>(st + b_off + nt * 1024);
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = ((wn * 64 + nt * 16) * 2 + (i & 3) * 8) ^ fx;
                const unsigned char* q = st + A_BYTES + krow * 256 + col;
                s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * 256);
                s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[nt].v = __builtin_bit_cast(bf16x8, w);
            }
        }
        if (!A_KS) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) af[mt].v = *reinterpret_cast<const bf16x8*>(st + a_off + mt * 1024);
        } else {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int col = ((wm * 128 + mt * 16) * 2 + (i & 3) * 8) ^ fx;
                const unsigned char* q = st + krow * 512 + col;
                s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * 512);
                s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[mt].v = __builtin_bit_cast(bf16x8, w);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) mma16(acc[mt][nt], bfr[nt], af[mt]);
        if (want_colsum) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) mma16(csum[mt], ones, af[mt]);
        }
        __builtin_amdgcn_s_setprio(0);
        if (t + 1 < nk) {
            if (dma) POLUS_VMCNT(6); else POLUS_VMCNT(0);   // tile t+1 landed
            POLUS_LGKMCNT0();
            __builtin_amdgcn_s_barrier();
        }
        stage = stage == 2 ? 0 : stage + 1;
    }

    // every wave is done with the operand stages (and no DMA is in flight): reuse LDS for the C staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (want_colsum && g == 0) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = m0 + wm * 128 + mt * 16 + i;
            if (m < p.M) p.colsum_a[(long)split * p.M + m] = csum[mt][0];
        }
    }
    GemmArgs q = p;
    q.C = static_cast<TC*>(p.C) + (long)split * p.c_split_stride;
    if (MODE >= 0) epilogue_wave<TC, 64, DROP, MODE < 0 ? 0 : MODE, false>(q, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wid * 8704);
    else epilogue_wave_128x64_lds<TC, DROP>(q, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wid * 8704);
}

template <typename TC, bool A_KS, bool B_KS, bool DROP, int MODE = -1>
__global__ __launch_bounds__(NTHR, 2) void gemm_ring_kernel(GemmArgs p) {
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    ring_body<TC, A_KS, B_KS, DROP, MODE>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// Several problems with the same contraction length in one launch (the dW = dY^T X of all four
// Dense layers of an encoder layer): blockIdx.x runs over the concatenated (split, tile) lists of
// the problems (each tile list padded to a multiple of 8 so that workgroup index mod 8 stays the
// XCD).  One problem alone leaves most of the 512 workgroup slots empty unless K is cut into many
// splits, each of which costs an f32 slab to write and re-read; together they need 2-3, chosen per
// problem so that the launch is one full round of slots.
struct RingGroupArgs {
    GemmArgs p[POLUS_MAX_GROUP];
    int wg0[POLUS_MAX_GROUP + 1];     // first workgroup of each problem, ascending
    int tiles[POLUS_MAX_GROUP];       // real tiles of each problem
    int tpad[POLUS_MAX_GROUP];        // tiles rounded up to a multiple of 8
    int n;
};
template <typename TC, bool A_KS, bool B_KS>
__global__ __launch_bounds__(NTHR, 2) void gemm_ring_grouped_kernel(RingGroupArgs ga) {
    const int b = blockIdx.x;
    int q = 0;
#pragma unroll
    for (int k = 1; k < POLUS_MAX_GROUP; ++k)
        if (k < ga.n && b >= ga.wg0[k]) q = k;
    const int rel = b - ga.wg0[q];
    const int split = rel / ga.tpad[q], wg = rel - split * ga.tpad[q];
    if (wg >= ga.tiles[q]) return;          // padding workgroup
    // ga.p[q] with a run-time q would put the by-value argument array into scratch: read the
    // chosen problem straight out of the kernarg segment (scalar loads) into a local copy instead
    typedef const __attribute__((address_space(4))) unsigned char* karg_t;
    karg_t ka = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    GemmArgs P;
    {
        static_assert(sizeof(GemmArgs) % 4 == 0, "GemmArgs is copied word by word");
        const __attribute__((address_space(4))) uint32_t* src =
            reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(ka + offsetof(RingGroupArgs, p) + (size_t)q * sizeof(GemmArgs));
        uint32_t* dst = reinterpret_cast<uint32_t*>(&P);
#pragma unroll
        for (int w = 0; w < (int)(sizeof(GemmArgs) / 4); ++w) dst[w] = src[w];
        // the word copy hides that these are global pointers (flat loads/stores otherwise)
typedef __attribute__((address_space(1))) void gvoid_t;
        typedef __attribute__((address_space(1))) float gfloat_t;
        P.A = (const void*)(const gvoid_t*)P.A; P.B = (const void*)(const gvoid_t*)P.B; P.C = (void*)(gvoid_t*)P.C;
        P.bias = (const float*)(const gfloat_t*)P.bias; P.resid = (const void*)(const gvoid_t*)P.resid;
        P.aux = (void*)(gvoid_t*)P.aux; P.partial = (float*)(gfloat_t*)P.partial; P.colsum_a = (float*)(gfloat_t*)P.colsum_a;
    }
    ring_body<TC, A_KS, B_KS, false>(P, wg, ga.tiles[q], split);
}

template <typename TC, bool A_KS, bool B_KS, bool DROP = false, int MODE = -1>
int launch_ring(const GemmArgs& a, int splits, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_ring_kernel<TC, A_KS, B_KS, DROP, MODE>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_done = true;
    }
    const int tiles = ((a.M + TM - 1) / TM) * ((a.N + TN - 1) / TN);
    hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(NTHR), SMEM_BYTES, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm(ring 256x128)");
    return POLUS_OK;
}

template <typename TC>
int launch_layout(const GemmArgs& a, int a_ks, int b_ks, int splits, hipStream_t st) {
    if (!a_ks && !b_ks) return launch_ring<TC, false, false>(a, splits, st);
    if (!a_ks && b_ks) return launch_ring<TC, false, true>(a, splits, st);
    if (a_ks && b_ks) return launch_ring<TC, true, true>(a, splits, st);
    return launch_ring<TC, true, false>(a, splits, st);
}

}  // namespace

int polus_launch_gemm_ring_grouped_dw(const GemmArgs* probs, int n, const int* splits, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_ring_grouped_kernel<float, true, true>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_done = true;
    }
    RingGroupArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.n = n;
    int t0 = 0;
    for (int k = 0; k < n; ++k) {
        ga.p[k] = probs[k];
        ga.tiles[k] = ((probs[k].M + TM - 1) / TM) * ((probs[k].N + TN - 1) / TN);
        ga.tpad[k] = (ga.tiles[k] + 7) / 8 * 8;
        ga.wg0[k] = t0;
        t0 += ga.tpad[k] * splits[k];
    }
    ga.wg0[n] = t0;
    hipLaunchKernelGGL(kern, dim3(t0), dim3(NTHR), SMEM_BYTES, st, ga);
    POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(ring)");
    return POLUS_OK;
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 tile of the same ring: 4 waves 2(M) x 2(N), 64 x 64 per wave (4 x 4 MFMA tiles, 64 accumulator
// registers), LDS stage = A 128 rows x 64 B | B 128 rows x 64 B = 16 KiB, ring of 3 = 48 KiB: THREE workgroups per CU.
// Four LDS-DMA pieces per wave and K-step (A rows 32w.., B rows 32w..), `vmcnt(4)`.  Both operands K-contiguous,
// bf16 C, compile-time epilogue (pgemm::epilogue_wave_db with 4 m-tiles).  64 FLOP per filled byte against the
// 256 x 128 tile's 85, so it only pays where that tile leaves most of the chip idle: a few thousand tokens
// (M / 256 x N / 128 workgroups on 512 slots) -- twice the workgroups on 768 slots.
namespace r128 {
constexpr int TM_ = 128, TN_ = 128, A_BYTES_ = TM_ * 64, B_BYTES_ = TN_ * 64, STAGE_ = A_BYTES_ + B_BYTES_, SMEM_ = NSTAGE * STAGE_;
static_assert(4 * EpiDbCfg<64>::BYTES <= SMEM_, "the C staging buffers of the 4 waves reuse the operand stages");
}
template <bool DROP, int MODE>
__global__ __launch_bounds__(NTHR, 3) void gemm_ring128_kernel(GemmArgs p) {
    constexpr int TM_ = r128::TM_, TN_ = r128::TN_, A_BYTES_ = r128::A_BYTES_, STAGE_ = r128::STAGE_;
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = (p.N + TN_ - 1) / TN_;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * TM_, n0 = (wg % tiles_n) * TN_;
    const int K = p.K;
    const int nk = (K + TK - 1) / TK;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_chunk_ring);

    // per-lane DMA sources: wave w loads A rows 32w..32w+31 and B rows 32w..32w+31 (2 pieces each);
    // lane l of a piece covers row (l >> 2), LDS chunk (l & 3); same source-side swizzle as the 256-row tile
    const bf16_t* src[4];
    int kofs[4];
    bool rowok[4];
    int ldsofs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool isA = j < 2;
        const bf16_t* base = static_cast<const bf16_t*>(isA ? p.A : p.B);
        const long ld = isA ? p.lda : p.ldb;
        const int row = 32 * wid + 16 * (j & 1) + (lane >> 2);
        const int lc = (lane & 3) ^ pi4((row >> 2) & 3);
        const int gr = (isA ? m0 : n0) + row;
        rowok[j] = gr < (isA ? p.M : p.N);
        kofs[j] = lc * 8;
        src[j] = base + (long)gr * ld + lc * 8;
        ldsofs[j] = (isA ? 0 : A_BYTES_) + (32 * wid + 16 * (j & 1)) * 64;
    }
    auto issue = [&](int stage, int k0) {
        unsigned char* st = smem + stage * STAGE_;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16_t* sp = (rowok[j] && k0 + kofs[j] < K) ? src[j] + k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lds_void_t*)(st + ldsofs[j]), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fc = (g ^ pi4((i >> 2) & 3)) * 16;
    const int a_off = (wm * 64 + i) * 64 + fc;                // + mt * 1024
    const int b_off = A_BYTES_ + (wn * 64 + i) * 64 + fc;      // + nt * 1024

    issue(0, 0);
    if (nk > 1) { issue(1, TK); POLUS_VMCNT(4); } else { POLUS_VMCNT(0); }
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        const unsigned char* st = smem + stage * STAGE_;
        const bool dma = t + 2 < nk;
        if (dma) issue(stage == 0 ? 2 : stage - 1, (t + 2) * TK);     // the stage read in step t-1
        Frag<bf16_t> af[4], bfr[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bfr[nt].v = *reinterpret_cast<const bf16x8*>(st + b_off + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt].v = *reinterpret_cast<const bf16x8*>(st + a_off + mt * 1024);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) mma16(acc[mt][nt], bfr[nt], af[mt]);
        __builtin_amdgcn_s_setprio(0);
        if (t + 1 < nk) {
            if (dma) POLUS_VMCNT(4); else POLUS_VMCNT(0);             // tile t+1 landed
            POLUS_LGKMCNT0();
            __builtin_amdgcn_s_barrier();
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    // every wave is done with the operand stages and no DMA is in flight: LDS now stages C
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_wave_db<bf16_t, 64, DROP, MODE, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, smem + wid * EpiDbCfg<64>::BYTES);
}

template <bool DROP, int MODE>
int launch_ring128(const GemmArgs& a, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_ring128_kernel<DROP, MODE>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, r128::SMEM_));
        attr_done = true;
    }
    const int tiles = ((a.M + r128::TM_ - 1) / r128::TM_) * ((a.N + r128::TN_ - 1) / r128::TN_);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHR), r128::SMEM_, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm(ring 128 x 128)");
    return POLUS_OK;
}

int polus_launch_gemm_ring128(const GemmArgs& a, int mode, int drop, hipStream_t st) {
    switch (mode) {
        case 0: return launch_ring128<false, 0>(a, st);
        case 1: return launch_ring128<false, 1>(a, st);
        case 2: return drop ? launch_ring128<true, 2>(a, st) : launch_ring128<false, 2>(a, st);
        case 3: return launch_ring128<false, 3>(a, st);
    }
    return POLUS_ERR_INVALID;
}

int polus_launch_gemm_ring_dropout(const GemmArgs& a, hipStream_t st) {
    if (polus_gemm_epi_mode(a, 0, 1) == 2) return launch_ring<bf16_t, false, false, true, 2>(a, 1, st);
    return launch_ring<bf16_t, false, false, true>(a, 1, st);
}

int polus_launch_gemm_ring(const GemmArgs& a, int c_is_f32, int a_ks, int b_ks, int splits, hipStream_t st) {
    if (!c_is_f32 && !a_ks && !b_ks && splits == 1) {
        switch (polus_gemm_epi_mode(a, 0, 0)) {      // same epilogue classes as the persistent kernel
            case 0: return launch_ring<bf16_t, false, false, false, 0>(a, 1, st);
            case 1: return launch_ring<bf16_t, false, false, false, 1>(a, 1, st);
            case 2: return launch_ring<bf16_t, false, false, false, 2>(a, 1, st);
            case 3: return launch_ring<bf16_t, false, false, false, 3>(a, 1, st);
            default: break;
        }
    }
    return c_is_f32 ? launch_layout<float>(a, a_ks, b_ks, splits, st) : launch_layout<bf16_t>(a, a_ks, b_ks, splits, st);
}
