// Shared by the MFMA GEMM kernels (gemm.hip: 128x128 general; gemm_ring.hip / gemm_pp.hip / gemm_ppks.hip: bf16
// direct-to-LDS): launch arguments and the fused epilogues.
#pragma once
#include <type_traits>
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
    int epi_vec16;   // C / resid / aux / bias rows allow 16-byte accesses at 8-column granularity
    float drop_inv;       // 1/(1-p) when POLUS_GEMM_DROPOUT, mask index = m*N + n
    unsigned drop_thresh, drop_seed;
    float* colsum_a;      // ring kernel, A K-strided: per-split column sums of A, [splits][M] (bias gradient)
    long c_split_stride;  // elements between the C slabs of consecutive K-splits (ring kernel)
    int persist_all;  // POLUS_GEMM_PERSIST=2: the persistent form for every multi-round ping-pong launch (A/B)
    int persist; // gemm_pp.hip: > 0 = number of CUs for the persistent form of multi-round launches (POLUS_GEMM_PERSIST), 0 = one workgroup per tile
    int order;   // gemm_pp.hip: column tiles the concurrent tiles of one XCD span (0 = its run in row-major order)
    const PolusDyn* dyn;  // per-step scalars in device memory (graph replay) or null: kernels with a dropout epilogue
                          // replace drop_seed by polus_eff_seed(drop_seed, dyn) on entry
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
template <typename T, typename TC, bool DROP = false>
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
        apply_act_n<4, sizeof(T) == 2>(p.act, v);     // bf16 engine: polynomial GELU on every path (edge tiles too)
    }
    if (p.flags & POLUS_GEMM_ACT_BWD) {
        float u[4]; ld4x<T>(static_cast<const T*>(p.aux) + (long)m * p.ldaux + n, u, ev, nvalid);
        apply_act_grad_n<4, sizeof(T) == 2>(p.act, v, u);
    }
    if (DROP) {
        const unsigned base = (unsigned)m * (unsigned)p.N + (unsigned)n;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = polus_keep(p.drop_seed, base + r, p.drop_thresh) ? v[r] * p.drop_inv : 0.0f;
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

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// Same epilogue, staged through LDS so that HBM sees whole 128-byte lines: the accumulator
// layout (lane = one row, 4 columns; a store instruction touches 16 rows x 32 B) is turned into
// row-major (8 lanes = one 128-byte row segment of 64 columns, 16 B per lane) through a
// wave-private 32-row x 64-column f32 buffer (272-byte rows: conflict-free b128 writes).  LDS
// operations of one wave execute in order, so no barrier is involved.  `lds` must point to
// 8704 bytes owned by this wave; the caller guarantees nobody still reads operand tiles there.
template <typename TC, bool DROP = false>
__device__ __forceinline__ void epilogue_wave_128x64_lds(const GemmArgs& p, f32x4 (&acc)[8][4], int mb, int nb,
                                                         int lane, unsigned char* lds) {
    typedef bf16_t T;
    const int i = lane & 15, g = lane >> 4;
    const bool interior = p.epi_vec16 && !p.partial && (mb + 128 <= p.M) && (nb + 64 <= p.N);
    if (!interior) {
#pragma clang loop unroll(full)
        for (int mt = 0; mt < 8; ++mt)
#pragma clang loop unroll(full)
            for (int nt = 0; nt < 4; ++nt)
                epilogue_tile<T, TC, DROP>(p, acc[mt][nt], mb + mt * 16 + i, nb + nt * 16 + 4 * g, 0);
        return;
    }
    constexpr int RS = 272;
    const int rrow = lane >> 3, cg = lane & 7;
    const int ncol = nb + 8 * cg;
    float bv[8];
    if (p.bias) { load4<float>(p.bias + ncol, *reinterpret_cast<float(*)[4]>(&bv[0])); load4<float>(p.bias + ncol + 4, *reinterpret_cast<float(*)[4]>(&bv[4])); }
    else {
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[r] = 0.f;
    }
    const bool act_fwd = p.flags & POLUS_GEMM_ACT_FWD, act_bwd = p.flags & POLUS_GEMM_ACT_BWD;
    const bool accum = p.flags & POLUS_GEMM_ACCUM_C;
    const T* resid = static_cast<const T*>(p.resid);
    T* aux = static_cast<T*>(p.aux);
    TC* C = static_cast<TC*>(p.C);
#pragma clang loop unroll(full)
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(lds + (a * 16 + i) * RS + (nt * 16 + 4 * g) * 4) = acc[2 * c + a][nt];
#pragma clang loop unroll(full)
      for (int ph = 0; ph < 4; ph += 1) {
        bf16x8_t rr[1], uu[1];
        float oo[1][8];
#pragma unroll
        for (int pq = 0; pq < 1; ++pq) {
            const int ps = pq;
            const long m = mb + c * 32 + (ph + pq) * 8 + rrow;
            if (resid) rr[ps] = *reinterpret_cast<const bf16x8_t*>(resid + m * p.ldr + ncol);
            if (act_bwd) uu[ps] = *reinterpret_cast<const bf16x8_t*>(aux + m * p.ldaux + ncol);
            if (accum) {
                if (sizeof(TC) == 4) {
                    load4<float>(reinterpret_cast<const float*>(C) + m * p.ldc + ncol, *reinterpret_cast<float(*)[4]>(&oo[ps][0]));
                    load4<float>(reinterpret_cast<const float*>(C) + m * p.ldc + ncol + 4, *reinterpret_cast<float(*)[4]>(&oo[ps][4]));
                } else {
                    bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const T*>(C) + m * p.ldc + ncol);
#pragma unroll
                    for (int r = 0; r < 8; ++r) oo[ps][r] = (float)t[r];
                }
            }
        }
#pragma unroll
        for (int pq = 0; pq < 1; ++pq) {
            const int ps = pq;
            const long m = mb + c * 32 + (ph + pq) * 8 + rrow;
            const unsigned char* src = lds + ((ph + pq) * 8 + rrow) * RS + cg * 32;
            f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = v[r] * p.alpha + bv[r];
            if (act_fwd) {
                if (aux) {
                    bf16x8_t t;
#pragma unroll
                    for (int r = 0; r < 8; ++r) t[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x8_t*>(aux + m * p.ldaux + ncol) = t;
                }
                apply_act_n<8, true>(p.act, v);
            }
            if (act_bwd) {
                float u[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) u[r] = (float)uu[ps][r];
                apply_act_grad_n<8, true>(p.act, v, u);
            }
            if (DROP) {
                const unsigned base = (unsigned)m * (unsigned)p.N + (unsigned)ncol;
                polus_dropout_run<8>(v, p.drop_seed, base, p.drop_thresh, p.drop_inv, (p.N & 3) == 0);
            }
            if (resid) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += (float)rr[ps][r];
            }
            if (accum) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += oo[ps][r];
            }
            if (sizeof(TC) == 4) {
                float* dst = reinterpret_cast<float*>(C) + m * p.ldc + ncol;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                bf16x8_t t;
#pragma unroll
                for (int r = 0; r < 8; ++r) t[r] = (bf16_t)v[r];
                *reinterpret_cast<bf16x8_t*>(reinterpret_cast<T*>(C) + m * p.ldc + ncol) = t;
            }
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------
// (K-contiguous operands, bf16 C) gemm_ring.hip.
template <int WN> struct EpiCfg {
    static constexpr int RS = WN * 4 + 16;            // staging row stride (f32 row of the wave + pad)
    static constexpr int BYTES = 9 * RS;              // half an m-tile (8 rows) + one dump row
    static constexpr int SLOTS = 8 * (WN / 8);        // 8 rows x eight-column chunks, one per lane
    static constexpr int PASSES = (SLOTS + 63) / 64;
};
// Pins the point where an accumulator tile is read: the copy out of the accumulator registers
// cannot be hoisted above this (otherwise all copies pile up at the top of the epilogue and spill).
template <bool AGPR> __device__ __forceinline__ f32x4 acc_take(f32x4& acc) {
    if (AGPR) asm volatile("" : "+a"(acc) :: "memory");
    else asm volatile("" : "+v"(acc) :: "memory");
    return acc;
}
// ---------------------------------------------------------------------------------------------
// Epilogue of one wave: 128 rows x WN columns, one 16-row m-tile at a time through a
// wave-private f32 staging buffer so that global accesses are 16 B per lane on whole rows.
// Same operation order as pgemm::epilogue_tile.
// MODE (compile time, so that variants without loads carry no vmcnt waits between their stores --
// vmcnt retires in order, a wait for a residual load would also wait for every older C store):
//   0 = alpha/bias only, 1 = ACT_FWD (+ pre-activation to aux), 2 = residual (+ dropout), 3 = ACT_BWD (aux read)
// AGPR: the accumulators are pinned to AGPRs by inline-asm MFMAs, or live in VGPRs (gemm_ring.hip).
template <typename TC, int WN, bool DROP, int MODE, bool AGPR, int DEPTH_ = 0>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& p, f32x4 (&acc)[8][WN / 16], int mb, int nb,
                                              int lane, unsigned char* lds) {
    typedef bf16_t T;
    constexpr int NT = WN / 16, RS = EpiCfg<WN>::RS, PASSES = EpiCfg<WN>::PASSES, CPR = 2 * NT;  // 8-col chunks per row
    const int i = lane & 15, g = lane >> 4;
    const bool interior = p.epi_vec16 && (mb + 128 <= p.M) && (nb + WN <= p.N);
    if (!interior) {
#pragma clang loop unroll(full)
        for (int mt = 0; mt < 8; ++mt)
#pragma clang loop unroll(full)
            for (int nt = 0; nt < NT; ++nt)
                epilogue_tile<T, TC, DROP>(p, acc_take<AGPR>(acc[mt][nt]), mb + mt * 16 + i, nb + nt * 16 + 4 * g, 0);
        return;
    }
    constexpr bool act_fwd = MODE == 1, act_bwd = MODE == 3, accum = false, has_resid = MODE == 2;
    const T* resid = static_cast<const T*>(p.resid);
    T* aux = static_cast<T*>(p.aux);
    TC* C = static_cast<TC*>(p.C);
    constexpr int SLOTS = EpiCfg<WN>::SLOTS;
    int prow[PASSES], pcol[PASSES];
    float bv[PASSES][8];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        // lanes past the last slot repeat an earlier slot (same value to the same address) instead of
        // being masked off: the epilogue stays free of divergent control flow (hipcc 7.2 moved the
        // staging writes of the following m-tile into such a branch: wrong values from those lanes)
        const int slot = (ps * 64 + lane) % SLOTS;
        prow[ps] = slot / CPR;
        pcol[ps] = 8 * (slot % CPR);
        if (p.bias) {
            load4<float>(p.bias + nb + pcol[ps], *reinterpret_cast<float(*)[4]>(&bv[ps][0]));
            load4<float>(p.bias + nb + pcol[ps] + 4, *reinterpret_cast<float(*)[4]>(&bv[ps][4]));
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) bv[ps][r] = 0.f;
        }
    }
    // residual / aux rows are fetched DEPTH m-tiles ahead, so the (in-order) wait for them only has
    // to get past the previous m-tiles' stores' *issue*, never their completion.  In the training step
    // these rows come from HBM (written a layer or a whole forward pass earlier): with VGPR
    // accumulators (ring kernel, 64 spare registers) the prefetch distance is 4 m-tiles, with AGPR
    // accumulators (persistent kernel, fragments of the next tile live) 1.
    constexpr int DEPTH = DEPTH_ > 0 ? DEPTH_ : (AGPR ? 1 : 4), NBUF = DEPTH + 1;     // DEPTH_: the persistent ping-pong kernel keeps the next tile's row offsets live and has room for 2
    bf16x8_t pre[NBUF][2][PASSES];
    auto fetch = [&](int mt, bf16x8_t (&dst)[2][PASSES]) {
        const T* base = has_resid ? resid : static_cast<const T*>(aux);
        const long ld = has_resid ? p.ldr : p.ldaux;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps)
                dst[h][ps] = *reinterpret_cast<const bf16x8_t*>(base + (long)(mb + mt * 16 + h * 8 + prow[ps]) * ld + nb + pcol[ps]);
    };
    if (has_resid || act_bwd) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, pre[d % NBUF]);
    }
#pragma clang loop unroll(full)
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 t[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) t[nt] = acc_take<AGPR>(acc[mt][nt]);
        if ((has_resid || act_bwd) && mt + DEPTH < 8) fetch(mt + DEPTH, pre[(mt + DEPTH) % NBUF]);
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {
            // rows 8h..8h+7 of the m-tile live in lanes with (i>>3) == h; the other lanes write to a
            // dump row instead of being masked off (hipcc 7.2 sinks the following reads and their
            // math into a divergent `if` here, leaving the masked-off lanes with stale registers)
            {
                unsigned char* wrow = lds + (((i >> 3) == h) ? (i & 7) : 8) * RS;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    *reinterpret_cast<f32x4*>(wrow + (nt * 16 + 4 * g) * 4) = t[nt];
            }
#pragma clang loop unroll(full)
            for (int ps = 0; ps < PASSES; ++ps) {
                const long m = mb + mt * 16 + h * 8 + prow[ps];
                const int ncol = nb + pcol[ps];
                const unsigned char* src = lds + prow[ps] * RS + pcol[ps] * 4;
                f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = v[r] * p.alpha + bv[ps][r];
                if (act_fwd) {
                    if (aux) {
                        bf16x8_t tt;
#pragma unroll
                        for (int r = 0; r < 8; ++r) tt[r] = (bf16_t)v[r];
                        // the pre-activation is only read again in the backward pass: streaming store
                        // (round 4: plain stores here, non-temporal loads of these rows in the backward pass and non-temporal C stores
                        // all measured within noise, profiles/r04_ab_cache_policy.txt)
                        __builtin_nontemporal_store(tt, reinterpret_cast<bf16x8_t*>(aux + m * p.ldaux + ncol));
                    }
                    apply_act_n<8, true>(p.act, v);
                }
                if (act_bwd) {
                    float u[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) u[r] = (float)pre[mt % NBUF][h][ps][r];
                    apply_act_grad_n<8, true>(p.act, v, u);
                }
                if (DROP) {
                    const unsigned base = (unsigned)m * (unsigned)p.N + (unsigned)ncol;
                    polus_dropout_run<8>(v, p.drop_seed, base, p.drop_thresh, p.drop_inv, (p.N & 3) == 0);
                }
                if (has_resid) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)pre[mt % NBUF][h][ps][r];
                }
                if (sizeof(TC) == 4) {
                    float* dst = reinterpret_cast<float*>(C) + m * p.ldc + ncol;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    bf16x8_t tt;
#pragma unroll
                    for (int r = 0; r < 8; ++r) tt[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x8_t*>(reinterpret_cast<T*>(C) + m * p.ldc + ncol) = tt;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Epilogue of one wave with TWO whole-m-tile staging buffers (gemm_pp.hip, whose operand stages are dead by then:
// 2 x 16 rows x RS bytes per wave).  epilogue_wave above pays two LDS round trips per half m-tile -- its next
// staging write has to wait for the reads of the rows it overwrites to have been *consumed* -- which at two waves
// per SIMD is most of its time (tools/pp_epi_probe.py: 5.6 us per 256 x 256 tile for 433 instructions).  Here the
// reads of m-tile mt are followed at once by the staging writes of m-tile mt + 2 into the same buffer: LDS executes
// the operations of one wave in order, so the write-after-read needs no wait, and the math of mt then runs with
// every LDS access of the next two m-tiles already in flight.  Same arithmetic, same operation order and the same
// global accesses as epilogue_wave<..., AGPR = false>: results are bit-identical to it.
template <int WN> struct EpiDbCfg {
    static constexpr int RS = EpiCfg<WN>::RS;
    static constexpr int BUF = 16 * RS;
    static constexpr int BYTES = 2 * BUF;
};
// MT = m-tiles of 16 rows the wave owns (8 in gemm_pp.hip, 4 in the 128 x 128 ring tile).
template <typename TC, int WN, bool DROP, int MODE, int MT = 8>
__device__ __forceinline__ void epilogue_wave_db(const GemmArgs& p, f32x4 (&acc)[MT][WN / 16], int mb, int nb,
                                                 int lane, unsigned char* lds) {
    typedef bf16_t T;
    constexpr int NT = WN / 16, RS = EpiDbCfg<WN>::RS, BUF = EpiDbCfg<WN>::BUF, PASSES = EpiCfg<WN>::PASSES, CPR = 2 * NT;
    constexpr int SLOTS = EpiCfg<WN>::SLOTS;
    const int i = lane & 15, g = lane >> 4;
    const bool interior = p.epi_vec16 && (mb + 16 * MT <= p.M) && (nb + WN <= p.N);
    if (!interior) {
#pragma clang loop unroll(full)
        for (int mt = 0; mt < MT; ++mt)
#pragma clang loop unroll(full)
            for (int nt = 0; nt < NT; ++nt)
                epilogue_tile<T, TC, DROP>(p, acc_take<false>(acc[mt][nt]), mb + mt * 16 + i, nb + nt * 16 + 4 * g, 0);
        return;
    }
    constexpr bool act_fwd = MODE == 1, act_bwd = MODE == 3, has_resid = MODE == 2;
    const T* resid = static_cast<const T*>(p.resid);
    T* aux = static_cast<T*>(p.aux);
    TC* C = static_cast<TC*>(p.C);
    int prow[PASSES], pcol[PASSES];
    float bv[PASSES][8];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int slot = (ps * 64 + lane) % SLOTS;     // lanes past the last slot repeat an earlier one (see epilogue_wave)
        prow[ps] = slot / CPR;
        pcol[ps] = 8 * (slot % CPR);
        if (p.bias) {
            load4<float>(p.bias + nb + pcol[ps], *reinterpret_cast<float(*)[4]>(&bv[ps][0]));
            load4<float>(p.bias + nb + pcol[ps] + 4, *reinterpret_cast<float(*)[4]>(&bv[ps][4]));
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) bv[ps][r] = 0.f;
        }
    }
    constexpr int DEPTH = MT < 4 ? MT : 4, NBUF = DEPTH + 1;         // residual / aux rows fetched DEPTH m-tiles ahead (see epilogue_wave)
    bf16x8_t pre[NBUF][2][PASSES];
    auto fetch = [&](int mt, bf16x8_t (&dst)[2][PASSES]) {
        const T* base = has_resid ? resid : static_cast<const T*>(aux);
        const long ld = has_resid ? p.ldr : p.ldaux;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps)
                dst[h][ps] = *reinterpret_cast<const bf16x8_t*>(base + (long)(mb + mt * 16 + h * 8 + prow[ps]) * ld + nb + pcol[ps]);
    };
    auto stage = [&](int mt) {                         // lane (i, g): row i, columns nt * 16 + 4 g .. of every n-tile
        unsigned char* wrow = lds + (mt & 1) * BUF + i * RS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<f32x4*>(wrow + (nt * 16 + 4 * g) * 4) = acc_take<false>(acc[mt][nt]);
    };
    if (has_resid || act_bwd) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, pre[d % NBUF]);
    }
    stage(0);
    stage(1);
#pragma clang loop unroll(full)
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 lo[2][PASSES], hi[2][PASSES];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const unsigned char* src = lds + (mt & 1) * BUF + (h * 8 + prow[ps]) * RS + pcol[ps] * 4;
                lo[h][ps] = *reinterpret_cast<const f32x4*>(src);
                hi[h][ps] = *reinterpret_cast<const f32x4*>(src + 16);
            }
        if (mt + 2 < MT) stage(mt + 2);                 // behind the reads above in the wave's LDS queue: no wait needed
        if ((has_resid || act_bwd) && mt + DEPTH < MT) fetch(mt + DEPTH, pre[(mt + DEPTH) % NBUF]);
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {
#pragma clang loop unroll(full)
            for (int ps = 0; ps < PASSES; ++ps) {
                const long m = mb + mt * 16 + h * 8 + prow[ps];
                const int ncol = nb + pcol[ps];
                float v[8] = {lo[h][ps][0], lo[h][ps][1], lo[h][ps][2], lo[h][ps][3], hi[h][ps][0], hi[h][ps][1], hi[h][ps][2], hi[h][ps][3]};
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = v[r] * p.alpha + bv[ps][r];
                if (act_fwd) {
                    if (aux) {
                        bf16x8_t tt;
#pragma unroll
                        for (int r = 0; r < 8; ++r) tt[r] = (bf16_t)v[r];
                        __builtin_nontemporal_store(tt, reinterpret_cast<bf16x8_t*>(aux + m * p.ldaux + ncol));
                    }
                    apply_act_n<8, true>(p.act, v);
                }
                if (act_bwd) {
                    float u[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) u[r] = (float)pre[mt % NBUF][h][ps][r];
                    apply_act_grad_n<8, true>(p.act, v, u);
                }
                if (DROP) {
                    const unsigned base = (unsigned)m * (unsigned)p.N + (unsigned)ncol;
                    polus_dropout_run<8>(v, p.drop_seed, base, p.drop_thresh, p.drop_inv, (p.N & 3) == 0);
                }
                if (has_resid) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)pre[mt % NBUF][h][ps][r];
                }
                if (sizeof(TC) == 4) {
                    float* dst = reinterpret_cast<float*>(C) + m * p.ldc + ncol;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    bf16x8_t tt;
#pragma unroll
                    for (int r = 0; r < 8; ++r) tt[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x8_t*>(reinterpret_cast<T*>(C) + m * p.ldc + ncol) = tt;
                }
            }
        }
    }
}


// XCD-aware bijective remap: blocks b and b+8 share an XCD (and its L2); give each XCD a
// contiguous run of tiles so that neighbours reuse the same A row panel.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// Tile of workgroup `bid` in a grid of `nwg` = tiles_m x tiles_n tiles, XCD-aware.  Every XCD owns a contiguous run of
// row-major tiles (xcd_remap).  order = 0: the run is walked in row-major order -- an XCD's ~32 concurrent tiles then
// span 2-3 row panels x ALL column tiles, so every XCD streams the whole weight matrix through its 4 MiB L2 in every
// round (FFN1: 4.7 MB of W per round, 150 MB fetched per launch against 30 MB of operands).  order = c > 0 (when the
// run is whole rows and c divides tiles_n): the run is walked in blocks of R rows x c columns, row-major inside a block,
// so the concurrent tiles of an XCD share c weight column panels and R activation row panels.
__device__ __forceinline__ void tile_of(int bid, int nwg, int tiles_n, int order, int& row, int& col) {
    const int per = nwg >> 3;
    if (order > 0 && (nwg & 7) == 0 && per % tiles_n == 0 && tiles_n % order == 0 && tiles_n > order) {
        const int xcd = bid & 7, j = bid >> 3, R = per / tiles_n, blk = R * order;
        const int b = j / blk, jj = j - b * blk;
        row = xcd * R + jj / order;
        col = b * order + jj % order;
        return;
    }
    const int wg = xcd_remap(bid, nwg);
    row = wg / tiles_n;
    col = wg - row * tiles_n;
}

}  // namespace pgemm

// gemm_ring.hip: same contract, 256x128 tile, two workgroups per CU.
// a_ks / b_ks: operand stored [K][rows]; splits > 1: blockIdx.y selects [y*k_per_split, ..) and C + y*c_split_stride.
int polus_launch_gemm_ring(const pgemm::GemmArgs& a, int c_is_f32, int a_ks, int b_ks, int splits, hipStream_t st);
// gemm_ring.hip: 128 x 128 tile, three workgroups per CU, both operands K-contiguous, bf16 C, mode from polus_gemm_epi_mode:
// for launches whose 256-row tiles would leave most of the chip idle (a few thousand tokens).
int polus_launch_gemm_ring128(const pgemm::GemmArgs& a, int mode, int drop, hipStream_t st);
// several dW problems (both operands K-strided, f32 C / slabs, same K and k_per_split) in one launch
#define POLUS_MAX_GROUP 8
int polus_launch_gemm_ring_grouped_dw(const pgemm::GemmArgs* probs, int n, const int* splits, hipStream_t st);
// dropout epilogue (POLUS_GEMM_DROPOUT): bf16 C, both operands K-contiguous only.
int polus_launch_gemm_ring_dropout(const pgemm::GemmArgs& a, hipStream_t st);
// gemm.hip: compile-time epilogue class of a launch (0 bias, 1 act fwd, 2 residual (+ dropout), 3 act bwd; -1: none fits)
int polus_gemm_epi_mode(const pgemm::GemmArgs& a, int c_is_f32, int drop);
// gemm_pp.hip: 256 x tn tile (tn = 256 or 192), 8 waves in two half-phase-staggered groups, one workgroup per
// CU, both operands K-contiguous, K % 64 == 0, bf16 C, mode from polus_gemm_epi_mode.
int polus_launch_gemm_pp(const pgemm::GemmArgs& a, int mode, int drop, int tn, hipStream_t st);
// gemm_ppks.hip: the grouped dW launch on 256 x 256 tiles (both operands K-strided, f32 C / slabs, K % 64 == 0),
// and the one-launch reduction of a group's slabs and bias-gradient partials.
int polus_ppks_tiles(int n_out, int n_in);
int polus_launch_gemm_ppks_grouped_dw(const pgemm::GemmArgs* probs, int n, const int* splits, hipStream_t st);
int polus_launch_dw_group_reduce(int n, const float* const* slabs, const float* const* cs, float* const* dW, float* const* db,
                                 const long* lddw, const int* n_out, const int* n_in, const int* splits, int accumulate,
                                 hipStream_t st);

// gemm_ppks.hip: the grouped dW launch with `base` regular K-slices per tile plus a stream-K remainder on the CUs the even
// split leaves idle.  polus_ppks_sk_plan answers 0 when the hybrid does not apply (then the even split runs).
struct PPKSSKPlan {
    int tiles[POLUS_MAX_GROUP], tile0[POLUS_MAX_GROUP + 1], unit0[POLUS_MAX_GROUP + 1];
    int n, base, kr, krem, R, q_units, rem_units, slots, grid, ttot;
};
int polus_ppks_sk_plan(const int* n_out, const int* n_in, int n, int T, int ncu, int delta, PPKSSKPlan* pl);
int polus_launch_gemm_ppks_sk(const pgemm::GemmArgs* probs, const PPKSSKPlan& pl, float* slabs, float* const* cs, hipStream_t st);
int polus_launch_dw_group_reduce_sk(const PPKSSKPlan& pl, const float* slabs, const float* const* cs, float* const* dW, float* const* db,
                                    const long* lddw, const int* n_out, const int* n_in, int accumulate, hipStream_t st);
