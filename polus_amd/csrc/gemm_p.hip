// bf16 MFMA GEMM, persistent, 256 x 192 tile, ONE 4-wave workgroup per CU (gfx950).
// C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous.  OPT-IN (POLUS_GEMM_P=1|2): since the ring
// kernel got compile-time epilogue modes the two tie in the training step; kept because it is the
// base for the larger-tile work and is covered by the GPU tests.
//
// Why this shape: the operand fill path (L2 -> LDS) tops out near 10 TB/s for the whole chip
// (profiles/README.md), so FLOP per filled byte is what sets the ceiling: 256x128 = 85 FLOP/B
// (gemm_ring.hip, ~850 TF), 256x192 = 110, 256x256 = 128.  A wave owns 128 x TN/2 of C: 8 x 6
// (or 8 x 8) MFMA 16x16x32 tiles = 192 (256) accumulator registers, which only fits with one wave
// per SIMD (512 registers, accumulators in AGPRs) -- so there is no second workgroup to hide
// latencies behind, and the kernel hides them itself:
//   * persistent: grid = #CUs; workgroup w walks tiles w, w+grid, ... and the LDS-DMA ring runs
//     over the FLATTENED (tile, k-step) sequence, so the next tile's first stages are in flight
//     while this tile's epilogue runs (no prologue bubble per tile);
//   * ring of NS = 5 stages (K-step 32, A 256 rows x 64 B | B TN rows x 64 B): step s reads next
//     step's fragments from stage s+1 while its MFMAs run on registers loaded in step s-1;
//     stages s+2 .. s+NS are in flight.  One `s_waitcnt vmcnt((NS-2)*ND)` + one raw s_barrier
//     per K-step (ND = DMA instructions per wave per stage).  (4 vs 5 stages measure the same:
//     the fill path is bandwidth-, not latency-bound.)
//   * LOADS retire in order among themselves, but stores may retire before older loads: a counted
//     wait may only count younger loads, so the first steps after an epilogue also wait for its C
//     stores (a wait that discounted the stores raced: a stage was read before it had landed).
// TN = 192 divides every Dense width of BERT-base evenly AND makes tiles a multiple of 256
// at T = 16384 (N = 768: 256 tiles, 2304: 768, 3072: 1024) -- no tail round.
// Same swizzle as gemm_ring.hip: LDS chunk pc of row r holds K-chunk pc ^ pi((r>>2)&3).
#include <type_traits>
#include "gemm_common.h"

using namespace pgemm;

namespace {

constexpr int TM = 256, TK = 32, NTHR = 256, NS = 5;
constexpr int A_BYTES = TM * 64;

__device__ const uint4 g_zero_chunk_p[1] = {{0u, 0u, 0u, 0u}};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef const __attribute__((address_space(1))) bf16_t* gbf_t;

__device__ __forceinline__ int pi4(int q) { return (0x78 >> (2 * q)) & 3; }

// s_waitcnt through the builtin (not inline asm): the compiler's own wait-count tracking sees
// it, so it does not add conservative waits of its own in front of the MFMA block.
// gfx9 encoding: vmcnt = {imm[15:14], imm[3:0]}, expcnt = imm[6:4], lgkmcnt = imm[11:8].
#define POLUS_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define POLUS_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// MFMA with the accumulator pinned to AGPRs and updated in place: with 192 accumulator registers
// the register allocator otherwise splits their live ranges (copies through VGPRs every K-step
// and spilled fragments).  The hazard recogniser does not see inside inline asm, so the callers
// put explicit s_nop between the last MFMA and the first accumulator read.
__device__ __forceinline__ void mma16_agpr(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// first K-step of a tile: C input is the inline constant 0 (no accumulator zeroing pass)
__device__ __forceinline__ void mma16_agpr_zero(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
template <int TN> struct Cfg {
    static constexpr int NT = TN / 32;            // n-tiles per wave (wave = 128 x TN/2)
    static constexpr int NB = TN / 64;            // B DMA units (16 rows x 64 B) per wave per stage
    static constexpr int ND = 4 + NB;             // DMA instructions per wave per stage
    static constexpr int STAGE = A_BYTES + TN * 64;
    static constexpr int EPI = EpiCfg<TN / 2>::BYTES;   // wave-private epilogue staging
    static constexpr int SMEM = NS * STAGE + 4 * EPI;
};

template <typename TC, int TN, bool DROP, int MODE, int ABL = 0>
__global__ __launch_bounds__(NTHR, 1) void gemm_p_kernel(GemmArgs p) {
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    typedef Cfg<TN> CF;
    constexpr int NT = CF::NT, NB = CF::NB, ND = CF::ND, STAGE = CF::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = (p.N + TN - 1) / TN;
    const int tiles = ((p.M + TM - 1) / TM) * tiles_n;
    const int first = xcd_remap(blockIdx.x, gridDim.x);
    if (first >= tiles) return;
    const int ntl = (tiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nk = p.K / TK;            // host guarantees K % 64 == 0
    const int G = ntl * nk;             // stages in this workgroup's flattened (tile, k-step) stream

    // ---- DMA side: wave w loads A rows 64w + 16j (j < 4) and B rows (TN/4)w + 16j (j < NB);
    // lane l of a unit covers row (l>>2), LDS chunk (l&3) <- K-chunk (l&3) ^ pi((l>>4)&3)
    const int drow = lane >> 2;
    const int lc8 = ((lane & 3) ^ pi4((lane >> 4) & 3)) * 8;
    const gbf_t zero = (gbf_t) reinterpret_cast<const bf16_t*>(g_zero_chunk_p);
    gbf_t src[ND];
    bool ok[ND];
    auto set_dma_tile = [&](int tile) {
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const bool isA = j < 4;
            const int row = isA ? tm * TM + 64 * wid + 16 * j + drow : tn * TN + (TN / 4) * wid + 16 * (j - 4) + drow;
            ok[j] = row < (isA ? p.M : p.N);
            const bf16_t* base = static_cast<const bf16_t*>(isA ? p.A : p.B);
            src[j] = (gbf_t)(base + (long)row * (isA ? p.lda : p.ldb) + lc8);
        }
    };
    int dtile = first, dk = 0, dleft = G, islot = 0;
    set_dma_tile(dtile);
    // one DMA instruction of the stage being issued (j = 0..ND-1); `finish_stage` advances the stream
    // diagnostics (POLUS_GEMM_ABLATE, compile-time variants): 1 = no in-loop DMA, 2 = no MFMA, 4 = no epilogue
    constexpr bool ab_nodma = ABL & 1, ab_nomma = ABL & 2, ab_noepi = ABL & 4;
    auto issue_one = [&](int j) {
        if (dleft <= 0 || ab_nodma) return;
        unsigned char* st = smem + islot * STAGE;
        const int lofs = j < 4 ? (64 * wid + 16 * j) * 64 : A_BYTES + ((TN / 4) * wid + 16 * (j - 4)) * 64;
        gbf_t sp = ok[j] ? src[j] + dk * TK : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)sp, (lds_void_t*)(st + lofs), 16, 0, 0);
    };
    auto finish_stage = [&]() {
        if (dleft <= 0) return;
        --dleft;
        islot = islot + 1 == NS ? 0 : islot + 1;
        if (++dk == nk) {
            dk = 0;
            dtile += gridDim.x;
            if (dleft > 0) set_dma_tile(dtile);
        }
    };
    auto issue_stage = [&]() {
#pragma unroll
        for (int j = 0; j < ND; ++j) issue_one(j);
        finish_stage();
    };

    // ---- MFMA side
    const int fc = (g ^ pi4((i >> 2) & 3)) * 16;
    const int a_off = (wm * 128 + i) * 64 + fc;                     // + mt * 1024
    const int b_off = A_BYTES + (wn * (TN / 2) + i) * 64 + fc;      // + nt * 1024

    f32x4 acc[8][NT];

    // prologue: stages 0..3 in flight, stage 0 landed, its fragments in registers
#pragma unroll
    for (int q = 0; q < NS; ++q) issue_stage();
    if (G >= NS) POLUS_WAIT_VM((NS - 1) * ND); else POLUS_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    Frag<bf16_t> fa0[8], fb0[NT], fa1[8], fb1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) fb0[nt].v = *reinterpret_cast<const bf16x8*>(smem + b_off + nt * 1024);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) fa0[mt].v = *reinterpret_cast<const bf16x8*>(smem + a_off + mt * 1024);

    int s = 0;        // global step
    int rslot = 1;    // ring slot of stage s+1
    constexpr int W_NORMAL = (NS - 2) * ND;
    // One K-step.  MFMAs run on (ca, cb); between them: the 7/8 DMA instructions of stage s+4 and
    // the reads of stage s+1's fragments into (na, nb), all in the first five m-rows so that the
    // LDS latency is covered by the remaining MFMAs before the next step's lgkmcnt(0).
    auto step = [&](auto first_tag, Frag<bf16_t> (&ca)[8], Frag<bf16_t> (&cb)[NT], Frag<bf16_t> (&na)[8], Frag<bf16_t> (&nb)[NT]) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int rem = G - 1 - s;
        const bool more = rem >= 1;
        if (more) {
            // stage s+1 landed (younger: s+2, s+3 when they exist), everyone done reading stage s
            // Loads retire in order among themselves, but stores may retire before older loads, so
            // the count can only be the number of younger LOADS: the first steps after an epilogue
            // also wait for its C stores (measured: ~1 us per tile).
            if (rem >= NS - 1) POLUS_WAIT_VM(W_NORMAL);
            else if (rem == 3) POLUS_WAIT_VM(2 * ND);
            else if (rem == 2) POLUS_WAIT_VM(ND);
            else POLUS_WAIT_VM(0);
            POLUS_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
        }
        const unsigned char* st = smem + rslot * STAGE;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (FIRST) mma16_agpr_zero(acc[mt][nt], cb[nt].v, ca[mt].v);
                else if (!ab_nomma) mma16_agpr(acc[mt][nt], cb[nt].v, ca[mt].v);
                else asm volatile("" :: "v"(cb[nt].v), "v"(ca[mt].v));
            }
            if (more) {
                // stage s+NS -> the slot whose fragments were read in step s-1: two DMA instructions per m-row
                if (2 * mt < ND) issue_one(2 * mt);
                if (2 * mt + 1 < ND) issue_one(2 * mt + 1);
                if (mt == 3) finish_stage();
                if (mt >= 1 && mt <= 2) {
#pragma unroll
                    for (int q = 0; q < NT / 2; ++q) {
                        const int nt = (mt - 1) * (NT / 2) + q;
                        nb[nt].v = *reinterpret_cast<const bf16x8*>(st + b_off + nt * 1024);
                    }
                }
                if (mt >= 2 && mt <= 5) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int m2 = (mt - 2) * 2 + q;
                        na[m2].v = *reinterpret_cast<const bf16x8*>(st + a_off + m2 * 1024);
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        ++s;
        rslot = rslot + 1 == NS ? 0 : rslot + 1;
    };

    unsigned char* my_epi = smem + NS * STAGE + wid * CF::EPI;
    int tile = first;
    for (int j = 0; j < ntl; ++j, tile += gridDim.x) {
        step(std::true_type{}, fa0, fb0, fa1, fb1);
        step(std::false_type{}, fa1, fb1, fa0, fb0);
        for (int kt = 2; kt < nk; kt += 2) {
            step(std::false_type{}, fa0, fb0, fa1, fb1);
            step(std::false_type{}, fa1, fb1, fa0, fb0);
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> v_accvgpr_read
        if (!ab_noepi) {
            epilogue_wave<TC, TN / 2, DROP, MODE, true>(p, acc, tm * TM + wm * 128, tn * TN + wn * (TN / 2), lane, my_epi);
        }
    }
}

template <typename TC, int TN, bool DROP, int MODE, int ABL = 0>
int launch_p(const GemmArgs& a, int ncu, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_p_kernel<TC, TN, DROP, MODE, ABL>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<TN>::SMEM));
        attr_done = true;
    }
    const int tiles = ((a.M + TM - 1) / TM) * ((a.N + TN - 1) / TN);
    const int grid = tiles < ncu ? tiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), Cfg<TN>::SMEM, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm(persistent 256xTN)");
    return POLUS_OK;
}

}  // namespace

// Epilogue mode of a call, or -1 when the combination is not built here (the caller stays on the
// ring kernel): see epilogue_wave.
int polus_gemm_p_mode(const GemmArgs& a, int c_is_f32, int drop) {
    if (c_is_f32 || (a.flags & POLUS_GEMM_ACCUM_C) || a.partial) return -1;
    const bool fwd = a.flags & POLUS_GEMM_ACT_FWD, bwd = a.flags & POLUS_GEMM_ACT_BWD;
    if (fwd && (bwd || a.resid || drop)) return -1;
    if (bwd && (a.resid || drop || !a.aux)) return -1;
    if (drop && !a.resid) return -1;
    if (fwd) return 1;
    if (bwd) return 3;
    if (a.resid) return 2;
    return 0;
}

// bf16 operands and C, both K-contiguous, K % 64 == 0, 16-byte aligned rows.
int polus_launch_gemm_p(const GemmArgs& a, int mode, int drop, int tn, int ncu, hipStream_t st) {
    if (tn != 192 || mode < 0) return POLUS_ERR_INVALID;
    if (a.ablate && mode == 0) {
        switch (a.ablate & 7) {
            case 1: return launch_p<bf16_t, 192, false, 0, 1>(a, ncu, st);
            case 2: return launch_p<bf16_t, 192, false, 0, 2>(a, ncu, st);
            case 4: return launch_p<bf16_t, 192, false, 0, 4>(a, ncu, st);
            case 5: return launch_p<bf16_t, 192, false, 0, 5>(a, ncu, st);
            case 6: return launch_p<bf16_t, 192, false, 0, 6>(a, ncu, st);
            default: break;
        }
    }
    switch (mode) {
        case 0: return launch_p<bf16_t, 192, false, 0>(a, ncu, st);
        case 1: return launch_p<bf16_t, 192, false, 1>(a, ncu, st);
        case 2: return drop ? launch_p<bf16_t, 192, true, 2>(a, ncu, st) : launch_p<bf16_t, 192, false, 2>(a, ncu, st);
        case 3: return launch_p<bf16_t, 192, false, 3>(a, ncu, st);
    }
    return POLUS_ERR_INVALID;
}
