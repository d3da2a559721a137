// bf16 MFMA GEMM, 256 x (256 | 192) tile, 8 waves in two half-phase-staggered groups (gfx950).
// C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous (forward Dense; dX through the transposed
// weight shadow), K % 64 == 0, bf16 C with a compile-time epilogue mode (pgemm::epilogue_wave).
//
// Why: the operand fill path (L2 -> LDS) delivers ~10 TB/s with the matrix pipe running, so FLOP per
// filled byte sets the ceiling (profiles/README.md): 256x128 (gemm_ring.hip) 85 -> ~0.9 PF,
// 256x192 110, 256x256 128 FLOP/B.  A tile that large needs all 8 waves of a CU in ONE workgroup, so
// nothing else is resident to cover its LDS reads and barriers; here the two wave groups cover each
// other instead.
//
// Waves 2(M) x 4(N): wave (wm, wn) owns rows wm*128.. (8 m-tiles of 16) x columns wn*WN.. (NT = 4 or 3
// n-tiles), 128 or 96 accumulator registers.  Group g = wm: waves 0-3 / 4-7, one of each on every
// SIMD.  A K-tile is 64 deep and is consumed in 4 phases of 2 m-tiles x NT n-tiles x 2 k-halves
// (16 / 12 MFMA 16x16x32).  A phase is   R: issue this phase's LDS-DMA, read its fragments
//                                         s_barrier (a)
//                                         M: s_waitcnt lgkmcnt(0), the MFMAs
//                                         s_barrier (b)
// and group 1 runs one barrier behind group 0, so between any two barriers one group's M section
// shares the SIMDs with the other group's R section: the matrix pipe always has a wave to take
// MFMAs from.  (B fragments are read once per K-tile in phase 0 and kept: 32 / 24 registers.)
//
// LDS: 2 stages x (A 256 rows x 128 B | B TN rows x 128 B) = 128 / 112 KiB, filled by
// global_load_lds_dwordx4 in 8 KiB pieces (one wave-instruction = 8 rows x 128 B, lane-linear, so the
// bank swizzle sits on the SOURCE address: LDS chunk c of row r holds K-chunk c ^ ((r >> 1) & 7);
// fragment reads apply the same XOR and every ds_read_b128 lane group hits 16 distinct 16-byte slots).
// A piece = the 32-row slab q of both wave groups (rows wm*128 + 32q ..), consumed by phase q;
// B pieces = 64 rows each, consumed from phase 0 on.
//
// Inside an R section the fragment reads come first and the LDS-DMA issue after them, and the section
// ends with lgkmcnt(0) BEFORE barrier (a): the LDS latency passes under the DMA issue instead of idling the
// matrix pipe at the head of the M section (tools/pp_ablate.py: with the wait behind the barrier the
// reads cost +10 us on a 54 us MFMA-only loop; reading the next phase's fragments underneath the MFMAs
// cost +20 us).  The region a phase consumes is therefore free one phase later and is refilled then
// with the data of the K-tile two ahead:
//     phase 0 of tile t: A slab 3 (and B piece 3) of tile t+1
//     phase P = 1..3   : A slab P-1 and B piece P-1 of tile t+2          (2 loads per wave and phase)
// The R sections run at raised priority (s_setprio 2): their ~20 instructions share the SIMD's issue port with
// the other group's MFMA stream, and left at equal priority they stretch past the M section they are
// meant to hide under (-8 % kernel time; priority on the M sections instead is the slower way round).
// i.e. every byte has >= 3 phases to land.  One counted wait per K-tile: `vmcnt(6)` at the end of phase
// 3's R section (everything issued up to phase 0 has landed: all of tile t+1); data is read in an R
// section that starts after the wait and a barrier every wave has passed.
#include <type_traits>
#include "gemm_common.h"

using namespace pgemm;

namespace {

constexpr int TM = 256, TK = 64, NTHR = 512;
constexpr int A_REGION = TM * 128;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int N> __device__ __forceinline__ void vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NT> struct PPCfg {
    static constexpr int WN = 16 * NT, TN = 4 * WN, NB = TN / 64;
    static constexpr int STAGE = A_REGION + TN * 128;
    static constexpr int SMEM = 2 * STAGE;
    static_assert(8 * ((EpiDbCfg<WN>::BYTES + 255) / 256 * 256) <= SMEM, "the C staging buffers of the 8 waves reuse the operand stages");
};

template <int NT, bool DROP, int MODE>
__global__ __launch_bounds__(NTHR, 2) void gemm_pp_kernel(GemmArgs p) {
    typedef PPCfg<NT> C;
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    constexpr int WN = C::WN, TN = C::TN, NB = C::NB, STAGE = C::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 2, wn = wid & 3;

    const int tiles_n = (p.N + TN - 1) / TN;
    int trow, tcol;
    tile_of(blockIdx.x, gridDim.x, tiles_n, p.order, trow, tcol);
    const int m0 = trow * TM, n0 = tcol * TN;
    const int nk = p.K / TK;

    // ---- LDS-DMA sources.  Rows past the edge are clamped (their accumulators are never stored).
    const int lrow = lane >> 3;
    const bf16_t* a_src;   // slab 0, tile 0; slab q adds 32 q rows, tile t adds 64 t elements
    const bf16_t* b_src;   // piece 0; piece b adds 64 b rows
    {
        const int ar = wm * 128 + 8 * wn + lrow;
        const int alc = (lane & 7) ^ ((ar >> 1) & 7);
        a_src = static_cast<const bf16_t*>(p.A) + alc * 8;
        const int br = 8 * wid + lrow;
        const int blc = (lane & 7) ^ ((br >> 1) & 7);
        b_src = static_cast<const bf16_t*>(p.B) + blc * 8;
    }
    long a_rowofs[4], b_rowofs[NB];
#pragma unroll
    for (int q = 0; q < 4; ++q) a_rowofs[q] = (long)min(m0 + wm * 128 + 32 * q + 8 * wn + lrow, p.M - 1) * p.lda;
#pragma unroll
    for (int b = 0; b < NB; ++b) b_rowofs[b] = (long)min(n0 + 64 * b + 8 * wid + lrow, p.N - 1) * p.ldb;
    const int a_dst = (wm * 128 + 8 * wn) * 128;          // + 32 q * 128
    const int b_dst = A_REGION + (8 * wid) * 128;         // + 64 b * 128
    auto load_a = [&](int tile, int q) {
        __builtin_amdgcn_global_load_lds((gptr_t)(a_src + a_rowofs[q] + (long)tile * TK),
                                         (lds_void_t*)(smem + (tile & 1) * STAGE + a_dst + q * 4096), 16, 0, 0);
    };
    auto load_b = [&](int tile, int b) {
        __builtin_amdgcn_global_load_lds((gptr_t)(b_src + b_rowofs[b] + (long)tile * TK),
                                         (lds_void_t*)(smem + (tile & 1) * STAGE + b_dst + b * 8192), 16, 0, 0);
    };

    // ---- fragment addresses: row R = base16 + i, K-chunk 4h + g at physical chunk ^ ((R >> 1) & 7)
    const int swz = (i >> 1) & 7;
    const int c0 = (g ^ swz) * 16, c1 = ((4 + g) ^ swz) * 16;
    const int a_off = (wm * 128 + i) * 128;                // + mt * 2048
    const int b_off = A_REGION + (wn * WN + i) * 128;      // + nt * 2048

    f32x4 acc[8][NT];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag<bf16_t> bfr[NT][2], af[2][2];         // af[m-tile of the phase][k-half]
    auto read_a = [&](int tile, int slab, int m) {
        const unsigned char* st = smem + (tile & 1) * STAGE + a_off + (2 * slab + m) * 2048;
        af[m][0].v = *reinterpret_cast<const bf16x8*>(st + c0);
        af[m][1].v = *reinterpret_cast<const bf16x8*>(st + c1);
    };
    auto read_b = [&](int tile, int nt) {
        const unsigned char* st = smem + (tile & 1) * STAGE + b_off + nt * 2048;
        bfr[nt][0].v = *reinterpret_cast<const bf16x8*>(st + c0);
        bfr[nt][1].v = *reinterpret_cast<const bf16x8*>(st + c1);
    };

    // ---- prologue: all of tile 0, then what phases 1-3 of "tile -1" would have issued for tile 1
#pragma unroll
    for (int q = 0; q < 4; ++q) load_a(0, q);
#pragma unroll
    for (int b = 0; b < NB; ++b) load_b(0, b);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { load_a(1, q); load_b(1, q); }
        vmcnt<6>();
    } else {
        vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier behind

    // TAIL: 0 = tiles t+1 and t+2 exist, 1 = only t+1, 2 = last tile
    auto phase = [&](auto P_, auto TAIL_, int t) {
        constexpr int P = decltype(P_)::value, TAIL = decltype(TAIL_)::value;
        // R: this phase's fragments first -- their LDS latency passes under the LDS-DMA issue below --
        // then the refill of what the previous phase's reads released
        __builtin_amdgcn_s_setprio(2);
        if (P == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) read_b(t, nt);
        }
        read_a(t, P, 0);
        read_a(t, P, 1);
        // (round 4: the same two loads issued BEHIND the barrier -- the R section then ends with the fragment reads alone -- is 1-6 %
        // slower on every shape, profiles/r04_ab_dma_behind_barrier.txt: the DMA issue is better hidden beside the other group's MFMAs)
        if (P == 0 && TAIL <= 1) { load_a(t + 1, 3); if (NB == 4) load_b(t + 1, 3); }
        if (P >= 1 && TAIL == 0) { load_a(t + 2, P - 1); load_b(t + 2, P - 1); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers: their LDS region is free
        if (P == 3) {                                         // LDS-DMA data of the next K-tile has landed
            if (TAIL == 0) vmcnt<6>(); else vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mma16(acc[2 * P + m][nt], bfr[nt][0], af[m][0]);
                mma16(acc[2 * P + m][nt], bfr[nt][1], af[m][1]);
            }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    auto ktile = [&](auto TAIL_, int t) {
        phase(std::integral_constant<int, 0>{}, TAIL_, t);
        phase(std::integral_constant<int, 1>{}, TAIL_, t);
        phase(std::integral_constant<int, 2>{}, TAIL_, t);
        phase(std::integral_constant<int, 3>{}, TAIL_, t);
    };
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) { ktile(std::integral_constant<int, 1>{}, t); ++t; }
    ktile(std::integral_constant<int, 2>{}, t);

    if (wm == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's last barrier (b)
    // every wave is done with the operand stages and no DMA is in flight: LDS now stages C
    // (two staging buffers per wave: bit-identical to the ring kernel's epilogue_wave, tests/test_kernels_gpu.py)
    epilogue_wave_db<bf16_t, WN, DROP, MODE>(p, acc, m0 + wm * 128, n0 + wn * WN, lane,
                                             smem + wid * ((EpiDbCfg<WN>::BYTES + 255) / 256 * 256));
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form for launches of several rounds (K = 768 with N = 2304 / 3072: three tiles per CU).  Same main loop;
// one workgroup per CU walks its tiles (those the hardware would have dealt to that CU: index + k x grid), and the
// operand prologue of the NEXT tile -- all of its K-tile 0 and three quarters of K-tile 1, what a fresh workgroup waits
// for before its first MFMA -- is issued BEFORE the epilogue of the current one: both operand stages are dead by then,
// so the 2-3 us of L2 / Infinity-Cache latency pass under the epilogue's stores instead of in front of the next tile.
// The epilogue therefore cannot stage C in the operand stages: it uses pgemm::epilogue_wave (one half-m-tile buffer per
// wave, 8 x 2.4 KiB behind the two stages) -- same arithmetic, same operation order, bit-identical results.
template <int NT> struct PPPCfg {
    static constexpr int WN = 16 * NT;
    static constexpr int EPI = (EpiCfg<WN>::BYTES + 255) / 256 * 256;
    static constexpr int SMEM = PPCfg<NT>::SMEM + 8 * EPI;
    static_assert(SMEM <= 160 * 1024, "two operand stages + the C staging of the 8 waves in one CU's LDS");
};

// The persistent loop of one workgroup.  `vb` of `vgrid`: this workgroup's index among those that share its tile list
// (vb & 7 = blockIdx.x & 7, the XCD); the list = the 256 x TN tiles of columns [n_lo, n_hi) of C, n_hi <= p.N (rows past
// p.M and columns past p.N are clamped / skipped as in gemm_pp_kernel).
template <int NT, bool DROP, int MODE>
__device__ __forceinline__ void ppp_run(GemmArgs& p, const int vb, const int vgrid, const int n_lo, const int n_hi) {
    typedef PPCfg<NT> C;
    constexpr int WN = C::WN, TN = C::TN, NB = C::NB, STAGE = C::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 2, wn = wid & 3;
    const int tiles_n = (n_hi - n_lo + TN - 1) / TN;
    const int ntiles = ((p.M + TM - 1) / TM) * tiles_n;
    const int nk = p.K / TK;

    const int lrow = lane >> 3;
    const bf16_t* a_src;
    const bf16_t* b_src;
    {
        const int ar = wm * 128 + 8 * wn + lrow;
        a_src = static_cast<const bf16_t*>(p.A) + ((lane & 7) ^ ((ar >> 1) & 7)) * 8;
        const int br = 8 * wid + lrow;
        b_src = static_cast<const bf16_t*>(p.B) + ((lane & 7) ^ ((br >> 1) & 7)) * 8;
    }
    long a_rowofs[4], b_rowofs[NB];
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a_rowofs[q] = (long)min(m0 + wm * 128 + 32 * q + 8 * wn + lrow, p.M - 1) * p.lda;
#pragma unroll
        for (int b = 0; b < NB; ++b) b_rowofs[b] = (long)min(n0 + 64 * b + 8 * wid + lrow, p.N - 1) * p.ldb;
    };
    const int a_dst = (wm * 128 + 8 * wn) * 128;
    const int b_dst = A_REGION + (8 * wid) * 128;
    auto load_a = [&](int tile, int q) {
        __builtin_amdgcn_global_load_lds((gptr_t)(a_src + a_rowofs[q] + (long)tile * TK),
                                         (lds_void_t*)(smem + (tile & 1) * STAGE + a_dst + q * 4096), 16, 0, 0);
    };
    auto load_b = [&](int tile, int b) {
        __builtin_amdgcn_global_load_lds((gptr_t)(b_src + b_rowofs[b] + (long)tile * TK),
                                         (lds_void_t*)(smem + (tile & 1) * STAGE + b_dst + b * 8192), 16, 0, 0);
    };
    auto issue_prologue = [&]() {          // all of K-tile 0, then what phases 1-3 of "K-tile -1" would have issued for K-tile 1
#pragma unroll
        for (int q = 0; q < 4; ++q) load_a(0, q);
#pragma unroll
        for (int b = 0; b < NB; ++b) load_b(0, b);
        if (nk > 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { load_a(1, q); load_b(1, q); }
        }
    };
    const int swz = (i >> 1) & 7;
    const int c0 = (g ^ swz) * 16, c1 = ((4 + g) ^ swz) * 16;
    const int a_off = (wm * 128 + i) * 128;
    const int b_off = A_REGION + (wn * WN + i) * 128;

    int it = vb, trow, tcol;
    bool prev_interior = false;
    if (it >= ntiles) return;
    tile_of(it, ntiles, tiles_n, p.order, trow, tcol);
    int m0 = trow * TM, n0 = n_lo + tcol * TN;
    set_tile(m0, n0);
    issue_prologue();

    for (; it < ntiles;) {
        f32x4 acc[8][NT];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Frag<bf16_t> bfr[NT][2], af[2][2];
        auto read_a = [&](int tile, int slab, int m) {
            const unsigned char* st = smem + (tile & 1) * STAGE + a_off + (2 * slab + m) * 2048;
            af[m][0].v = *reinterpret_cast<const bf16x8*>(st + c0);
            af[m][1].v = *reinterpret_cast<const bf16x8*>(st + c1);
        };
        auto read_b = [&](int tile, int nt) {
            const unsigned char* st = smem + (tile & 1) * STAGE + b_off + nt * 2048;
            bfr[nt][0].v = *reinterpret_cast<const bf16x8*>(st + c0);
            bfr[nt][1].v = *reinterpret_cast<const bf16x8*>(st + c1);
        };
        // The prologue of this tile was issued before the previous tile's epilogue, whose stores are YOUNGER.  vmcnt retires in
        // order, so `vmcnt(n)` covers the prologue exactly when the epilogue issued AT LEAST n vector-memory operations behind
        // it: an interior epilogue issues 16 x PASSES C stores per wave, and as many pre-activation stores on top when MODE 1
        // is given an aux buffer (the training forward; inference and the frozen encoders pass none).  The bias / aux /
        // residual loads only add to that.  A larger n than the operations really issued would reach INTO the prologue and let
        // the first MFMAs read LDS before the DMA landed, so n is the guaranteed minimum of the launch, never more; an edge
        // tile's epilogue has another count: full drain.
        constexpr int C_STORES = 16 * EpiCfg<WN>::PASSES;
        static_assert(2 * C_STORES <= 63, "vmcnt is a 6-bit counter");
        if (it == vb) { if (nk > 1) vmcnt<6>(); else vmcnt<0>(); }
        else if (prev_interior) { if (MODE == 1 && p.aux) vmcnt<2 * C_STORES>(); else vmcnt<C_STORES>(); }
        else vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier behind

        auto phase = [&](auto P_, auto TAIL_, int t) {
            constexpr int P = decltype(P_)::value, TAIL = decltype(TAIL_)::value;
            __builtin_amdgcn_s_setprio(2);
            if (P == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) read_b(t, nt);
            }
            read_a(t, P, 0);
            read_a(t, P, 1);
            if (P == 0 && TAIL <= 1) { load_a(t + 1, 3); if (NB == 4) load_b(t + 1, 3); }
            if (P >= 1 && TAIL == 0) { load_a(t + 2, P - 1); load_b(t + 2, P - 1); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (P == 3) {
                if (TAIL == 0) vmcnt<6>(); else vmcnt<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    mma16(acc[2 * P + m][nt], bfr[nt][0], af[m][0]);
                    mma16(acc[2 * P + m][nt], bfr[nt][1], af[m][1]);
                }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        auto ktile = [&](auto TAIL_, int t) {
            phase(std::integral_constant<int, 0>{}, TAIL_, t);
            phase(std::integral_constant<int, 1>{}, TAIL_, t);
            phase(std::integral_constant<int, 2>{}, TAIL_, t);
            phase(std::integral_constant<int, 3>{}, TAIL_, t);
        };
        int t = 0;
        for (; t + 2 < nk; ++t) ktile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) { ktile(std::integral_constant<int, 1>{}, t); ++t; }
        ktile(std::integral_constant<int, 2>{}, t);
        if (wm == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's last barrier (b)

        // every wave is done with both operand stages and no DMA is in flight: the next tile's operands start now
        const int cm0 = m0, cn0 = n0;
        prev_interior = p.epi_vec16 && (cm0 + TM <= p.M) && (cn0 + TN <= p.N);
        const int nxt = it + vgrid;
        it = nxt;
        if (it < ntiles) {
            tile_of(it, ntiles, tiles_n, p.order, trow, tcol);
            m0 = trow * TM; n0 = n_lo + tcol * TN;
            set_tile(m0, n0);
            issue_prologue();
        }
        epilogue_wave<bf16_t, WN, DROP, MODE, false, (NT == 4 ? (MODE == 3 || DROP ? 1 : 2) : 4)>(
            p, acc, cm0 + wm * 128, cn0 + wn * WN, lane, smem + 2 * STAGE + wid * PPPCfg<NT>::EPI);
    }
}

template <int NT, bool DROP, int MODE>
__global__ __launch_bounds__(NTHR, 2) void gemm_ppp_kernel(GemmArgs p) {
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    ppp_run<NT, DROP, MODE>(p, blockIdx.x, gridDim.x, 0, p.N);
}

template <int NT, bool DROP, int MODE>
int launch_ppp(const GemmArgs& a, int ncu, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_ppp_kernel<NT, DROP, MODE>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PPPCfg<NT>::SMEM));
        attr_done = true;
    }
    const int tiles = ((a.M + TM - 1) / TM) * ((a.N + PPCfg<NT>::TN - 1) / PPCfg<NT>::TN);
    hipLaunchKernelGGL(kern, dim3(tiles < ncu ? tiles : ncu), dim3(NTHR), PPPCfg<NT>::SMEM, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm(ping-pong 256-wide, persistent)");
    return POLUS_OK;
}
template <int NT>
int launch_ppp_mode(const GemmArgs& a, int mode, int drop, int ncu, hipStream_t st) {
    switch (mode) {
        case 0: return launch_ppp<NT, false, 0>(a, ncu, st);
        case 1: return launch_ppp<NT, false, 1>(a, ncu, st);
        case 2: return drop ? launch_ppp<NT, true, 2>(a, ncu, st) : launch_ppp<NT, false, 2>(a, ncu, st);
        case 3: return launch_ppp<NT, false, 3>(a, ncu, st);
    }
    return POLUS_ERR_INVALID;
}

template <int NT, bool DROP, int MODE>
int launch_pp(const GemmArgs& a, hipStream_t st) {
    typedef PPCfg<NT> C;
    static bool attr_done = false;
    auto kern = gemm_pp_kernel<NT, DROP, MODE>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
        attr_done = true;
    }
    const int tiles = ((a.M + TM - 1) / TM) * ((a.N + C::TN - 1) / C::TN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHR), C::SMEM, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm(ping-pong 256-wide)");
    return POLUS_OK;
}

template <int NT>
int launch_pp_mode(const GemmArgs& a, int mode, int drop, hipStream_t st) {
    switch (mode) {
        case 0: return launch_pp<NT, false, 0>(a, st);
        case 1: return launch_pp<NT, false, 1>(a, st);
        case 2: return drop ? launch_pp<NT, true, 2>(a, st) : launch_pp<NT, false, 2>(a, st);
        case 3: return launch_pp<NT, false, 3>(a, st);
    }
    return POLUS_ERR_INVALID;
}

}  // namespace

// tn = 256 or 192; mode from polus_gemm_epi_mode (>= 0); K % 64 == 0; bf16 C; 16-byte aligned rows.
int polus_launch_gemm_pp(const GemmArgs& a, int mode, int drop, int tn, hipStream_t st) {
    if (mode < 0 || a.K % TK != 0 || a.K < TK) return POLUS_ERR_INVALID;
    if (a.persist > 0) {
        // several rounds of tiles: one workgroup per CU walks them, the next tile's operand prologue under the epilogue
        const int tiles = ((a.M + TM - 1) / TM) * ((a.N + tn - 1) / tn);
        // measured (tools/pp_bench.py --ab POLUS_GEMM_PERSIST=0,1,2; bench.py --config c3 | c4 | c5): wins on the 256-wide launches --
        // the GELU ones of BERT-base (FFN1 forward, dU: -8 % from cold caches) and every multi-round launch of BERT-large (+0.5 %
        // on the c4 step) -- and is neutral to slightly negative on the 192-wide bias-only QKV launch of BERT-base (three exact
        // rounds, an epilogue too short to hide anything); POLUS_GEMM_PERSIST=2 forces it everywhere
        const bool wins = tn == 256 || a.persist_all;     // 192-wide launches lose with it (c5, 512 tiles of 256 x 192: 34.4 -> 34.8 ms/step)
        if (tiles > a.persist && wins) {
            if (tn == 256) return launch_ppp_mode<4>(a, mode, drop, a.persist, st);
            if (tn == 192) return launch_ppp_mode<3>(a, mode, drop, a.persist, st);
        }
    }
    if (tn == 256) return launch_pp_mode<4>(a, mode, drop, st);
    if (tn == 192) return launch_pp_mode<3>(a, mode, drop, st);
    return POLUS_ERR_INVALID;
}
