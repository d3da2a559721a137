// MFMA GEMM for the Dense layers of the BERT step and their gradients (gfx950).
//
// One 256-thread workgroup (4 waves, 2x2) computes a 128x128 tile of C; each wave owns
// 64x64 = 4x4 MFMA 16x16 tiles (64 accumulator registers).  A K-step is 128 bytes of K per
// row (64 bf16 / 32 f32).  Operands are staged HBM -> registers -> LDS (double-buffered,
// one barrier per K-step, the next tile's global loads in flight under the MFMAs).
//
// LDS images:
//   K-contiguous operand ([rows][K] in memory): [128 rows][128 B + 32 B pad] = 160-B rows;
//     fragments are ds_read_b128 of (row i, 16 B at k-group g) — 160 B puts the 16 rows of a
//     b128 lane group on 16 distinct 16-B slots.
//   K-strided operand ([K][rows] in memory: dY and X in dW = dY^T X, W in dX = dY W):
//     kept k-major exactly as loaded (coalesced 16-B chunks along the row index), bf16 rows
//     of 288 B with byte-bit-7 XOR for odd k-octets, f32 rows of 512 B with bit-6 XOR;
//     bf16 fragments come out of ds_read_b64_tr_b16 (hardware transpose), f32 ones from
//     ds_read_b32.
// The MFMA is issued as D[n][m] (B rows as the A operand) so that a lane ends up with four
// consecutive n of one output row: 8/16-byte epilogue accesses.
#include <stdlib.h>
#include "gemm_common.h"

using namespace pgemm;

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int KC_STRIDE = 160;
constexpr int OPERAND_BYTES = 128 * KC_STRIDE;  // 20480, >= every K-strided image
constexpr int SMEM_BYTES = 4 * OPERAND_BYTES;   // A0 B0 A1 B1

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int BK = 64, EPC = 8, NSUB = 2;
    static constexpr int KS_STRIDE = 288, KS_SWZ = 7, LOG_RCH = 4;
};
template <> struct Cfg<float> {
    static constexpr int BK = 32, EPC = 4, NSUB = 1;
    static constexpr int KS_STRIDE = 512, KS_SWZ = 6, LOG_RCH = 5;
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename T> union Chunk { uint4 u; u32x4_t w; T e[16 / sizeof(T)]; };

// ---- HBM -> registers: one operand tile = 1024 16-byte chunks, 4 per thread.
// Branch-free on purpose: a per-lane `if` around a global load makes hipcc wait vmcnt(0)
// at every join, which serialises the eight loads of a K-step.  Invalid chunks read a clamped
// (in-bounds) address and are zeroed by a select.  VEC = every 16-byte chunk is wholly valid
// or wholly invalid and 16-byte aligned (checked on the host); otherwise the element path
// loads each element with its own predicate (tiny heads only).
// 16 zero bytes in HBM: invalid chunks load from here, so no select has to consume the loaded
// value (a select right after the load would pull its s_waitcnt in front of the MFMAs).
__device__ const uint4 g_zero_chunk[1] = {{0u, 0u, 0u, 0u}};

template <typename T, bool VEC>
__device__ __forceinline__ uint4 load_chunk(const T* __restrict__ base, long off, bool ok, int nvalid) {
    constexpr int EPC = Cfg<T>::EPC;
    // both arms in the global address space: a generic-pointer select would turn into flat_load
    typedef const __attribute__((address_space(1))) u32x4_t* gp16_t;
    typedef const __attribute__((address_space(1))) T* gpe_t;
    Chunk<T> v;
    if (VEC) {
        gp16_t p = ok ? (gp16_t)(base + off) : (gp16_t)g_zero_chunk;
        v.w = *p;
    } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            gpe_t p = (ok && e < nvalid) ? (gpe_t)(base + off + e) : (gpe_t)g_zero_chunk;
            v.e[e] = *p;
        }
    }
    return v.u;
}
template <typename T, bool VEC>
__device__ __forceinline__ void gload_kc(uint4 (&r)[4], const T* __restrict__ base, long ld, int rows,
                                         int r0, int k0, int kend, int tid) {
    constexpr int EPC = Cfg<T>::EPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + NTHREADS * i;
        int gr = r0 + (c >> 3), gk = k0 + (c & 7) * EPC;
        r[i] = load_chunk<T, VEC>(base, (long)gr * ld + gk, gr < rows && gk < kend, kend - gk);
    }
}
template <typename T>
__device__ __forceinline__ void sstore_kc(unsigned char* tile, const uint4 (&r)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + NTHREADS * i;
        *reinterpret_cast<uint4*>(tile + (c >> 3) * KC_STRIDE + (c & 7) * 16) = r[i];
    }
}
template <typename T, bool VEC>
__device__ __forceinline__ void gload_ks(uint4 (&r)[4], const T* __restrict__ base, long ld, int rows,
                                         int r0, int k0, int kend, int tid) {
    constexpr int EPC = Cfg<T>::EPC, LOG = Cfg<T>::LOG_RCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + NTHREADS * i;
        int gk = k0 + (c >> LOG), gr = r0 + (c & ((1 << LOG) - 1)) * EPC;
        r[i] = load_chunk<T, VEC>(base, (long)gk * ld + gr, gk < kend && gr < rows, rows - gr);
    }
}
template <typename T>
__device__ __forceinline__ void sstore_ks(unsigned char* tile, const uint4 (&r)[4], int tid) {
    constexpr int LOG = Cfg<T>::LOG_RCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + NTHREADS * i;
        int krow = c >> LOG, rc = c & ((1 << LOG) - 1);
        int off = krow * Cfg<T>::KS_STRIDE + ((rc * 16) ^ (((krow >> 3) & 1) << Cfg<T>::KS_SWZ));
        *reinterpret_cast<uint4*>(tile + off) = r[i];
    }
}

// ---- LDS -> fragments
// K-contiguous image: tile row `row`, k-substep `sub`, lane group g
template <typename T>
__device__ __forceinline__ void frag_kc(Frag<T>& f, const unsigned char* tile, int row, int sub, int g) {
    frag_load_row(f, tile + row * KC_STRIDE + sub * (32 * (int)sizeof(T)) + g * (8 * (int)sizeof(T)));
}
// K-strided image: rows col0..col0+15 of the operand, k = sub*32 + 8g + j
__device__ __forceinline__ void frag_ks(Frag<bf16_t>& f, const unsigned char* tile, int col0, int sub, int i, int g) {
    int kb = sub * 32 + 8 * g + (i >> 2);
    int cb = (col0 + 4 * (i & 3)) * 2;
    int swz = (g & 1) << 7;  // ((k >> 3) & 1) for k = sub*32 + 8g + (0..7)
    s16x4 lo = lds_tr16(tile + kb * 288 + (cb ^ swz));
    s16x4 hi = lds_tr16(tile + (kb + 4) * 288 + (cb ^ swz));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ void frag_ks(Frag<float>& f, const unsigned char* tile, int col0, int sub, int i, int g) {
    int cb = ((col0 + i) * 4) ^ ((g & 1) << 6);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        f.v[j] = *reinterpret_cast<const float*>(tile + (8 * g + j) * 512 + cb);
}

template <typename T, bool A_KS, bool B_KS, typename TC, bool VEC, bool DROP>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs p) {
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = Cfg<T>::BK, NSUB = Cfg<T>::NSUB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const T* A = static_cast<const T*>(p.A);
    const T* B = static_cast<const T*>(p.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    auto gload = [&](int k0) {
        if (A_KS) gload_ks<T, VEC>(ra, A, p.lda, p.M, m0, k0, kend, tid);
        else      gload_kc<T, VEC>(ra, A, p.lda, p.M, m0, k0, kend, tid);
        if (B_KS) gload_ks<T, VEC>(rb, B, p.ldb, p.N, n0, k0, kend, tid);
        else      gload_kc<T, VEC>(rb, B, p.ldb, p.N, n0, k0, kend, tid);
    };
    auto sstore = [&](int buf) {
        unsigned char* ta = smem + buf * 2 * OPERAND_BYTES;
        unsigned char* tb = ta + OPERAND_BYTES;
        if (A_KS) sstore_ks<T>(ta, ra, tid); else sstore_kc<T>(ta, ra, tid);
        if (B_KS) sstore_ks<T>(tb, rb, tid); else sstore_kc<T>(tb, rb, tid);
    };

    if (nk > 0) { gload(kbeg); sstore(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
        const unsigned char* ta = smem + cur * 2 * OPERAND_BYTES;
        const unsigned char* tb = ta + OPERAND_BYTES;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            Frag<T> af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (A_KS) frag_ks(af[t], ta, wm * 64 + t * 16, sub, i, g);
                else      frag_kc<T>(af[t], ta, wm * 64 + t * 16 + i, sub, g);
                if (B_KS) frag_ks(bf[t], tb, wn * 64 + t * 16, sub, i, g);
                else      frag_kc<T>(bf[t], tb, wn * 64 + t * 16 + i, sub, g);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) mma16(acc[tm][tn], bf[tn], af[tm]);
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
            epilogue_tile<T, TC, DROP>(p, acc[tm][tn], m0 + wm * 64 + tm * 16 + i, n0 + wn * 64 + tn * 16 + 4 * g, blockIdx.z);
}

// order-fixed split-K reduction: C = alpha * sum_z slab[z] (+ bias) (+ C)
template <typename TC>
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int M, int N,
                                     TC* C, long ldc, float alpha, const float* bias, int accum) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)M * N;
    if (idx >= total) return;
    int m = idx / N, n = idx % N;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long)z * total + idx];
    s *= alpha;
    if (bias) s += bias[n];
    TC* c = C + (long)m * ldc + n;
    if (accum) s += to_f<TC>(*c);
    *c = from_f<TC>(s);
}

// order-fixed split-K reduction with the FULL epilogue of the unsplit kernels (bf16 engine, K-contiguous operands,
// bf16 C): a thread sums the slabs of four columns in slice order and hands them to pgemm::epilogue_tile -- bias,
// activation (+ pre-activation store), activation gradient, dropout, residual.  Lets a Dense GEMM that would fill
// less than half of the chip with 256-row tiles (a few thousand tokens, N = 768) run its K range in several slices.
template <bool DROP>
__global__ __launch_bounds__(256) void splitk_reduce_epi_kernel(GemmArgs p, const float* __restrict__ slabs, int splits) {
    if (DROP) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    const int nq = (p.N + 3) / 4;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)p.M * nq) return;
    const int m = (int)(q / nq), n = (int)(q % nq) * 4;
    const long total = (long)p.M * p.N;
    const float* s0 = slabs + (long)m * p.N + n;
    const int nvalid = p.N - n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        ld4x<float>(s0 + (long)z * total, v, (p.N & 3) == 0, nvalid);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    epilogue_tile<bf16_t, bf16_t, DROP>(p, acc, m, n, 0);
}

template <typename T, bool A_KS, bool B_KS, typename TC, bool VEC, bool DROP = false>
int launch_v(const GemmArgs& a, dim3 grid, hipStream_t st) {
    static bool attr_done = false;  // per instantiation
    auto kern = gemm_kernel<T, A_KS, B_KS, TC, VEC, DROP>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), SMEM_BYTES, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm");
    return POLUS_OK;
}

template <typename T, bool A_KS, bool B_KS, typename TC>
int launch(const GemmArgs& a, dim3 grid, hipStream_t st) {
    if (a.a_vec && a.b_vec) return launch_v<T, A_KS, B_KS, TC, true>(a, grid, st);
    return launch_v<T, A_KS, B_KS, TC, false>(a, grid, st);
}

template <typename T, typename TC>
int dispatch_layout(int al, int bl, const GemmArgs& a, dim3 grid, hipStream_t st) {
    if (al == POLUS_K_CONTIG && bl == POLUS_K_CONTIG) return launch<T, false, false, TC>(a, grid, st);
    if (al == POLUS_K_CONTIG && bl == POLUS_K_STRIDED) return launch<T, false, true, TC>(a, grid, st);
    if (al == POLUS_K_STRIDED && bl == POLUS_K_STRIDED) return launch<T, true, true, TC>(a, grid, st);
    return launch<T, true, false, TC>(a, grid, st);
}

}  // namespace

extern "C" size_t polus_gemm_workspace_bytes(int M, int N, int split_k) {
    if (split_k <= 1) return 0;
    return (size_t)split_k * (size_t)M * (size_t)N * sizeof(float);
}

// Epilogue class of a bf16-C launch for the kernels with compile-time epilogues (gemm_pp.hip, the 128 x 128 ring tile):
// 0 = alpha / bias, 1 = activation forward (+ pre-activation to aux), 2 = residual (+ dropout), 3 = activation backward
// (aux read); -1 = a combination they are not built for (the caller stays on the run-time epilogue of the ring kernel).
int polus_gemm_epi_mode(const GemmArgs& a, int c_is_f32, int drop) {
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

// The ping-pong kernel (gemm_pp.hip, one 8-wave workgroup per CU, 256 x 256 or 256 x 192 tile): more
// FLOP per byte filled into LDS than the ring kernel's 256 x 128, which is what bounds these GEMMs.
// Returns the tile width to use, 0 = stay on the ring kernel.  Needs K % 64 == 0, a compile-time
// epilogue mode and enough tiles to fill the chip: a launch is whole rounds of #CU tiles, so the
// tile width is chosen by the share of the last round that is busy (256 is ~10 % faster per tile).
static int pp_tile(int M, int N, int K, int mode, bool vec16) {
    const int sel = polus_cfg().gemm_pp;
    if (sel < 0 || mode < 0 || K % 64 != 0 || M < 256 || N < 192 || !vec16) return 0;
    if (sel == 256 || sel == 192) return sel;
    // CUs a launch can count on: all of them, minus what a concurrent gradient exchange holds (POLUS_GEMM_RESERVE_CUS: a
    // ping-pong workgroup fills a CU's registers and LDS, so a CU running an RCCL channel kernel takes no tile, and a
    // 256-tile launch that finds 240 free CUs runs two rounds)
    const int ncu = max(32, polus_num_cus() - polus_reserved_cus());
    const long tm = (M + 255) / 256;
    double best = 0.0; int best_tn = 0;
    for (int tn : {256, 192}) {
        const long tiles = tm * ((N + tn - 1) / tn);
        const long rounds = (tiles + ncu - 1) / ncu;
        const double useful = (double)M * N / ((double)rounds * ncu * 256.0 * tn);   // busy share incl. edge waste
        const double score = useful * (tn == 256 ? 1.10 : 1.0);
        if (score > best) { best = score; best_tn = tn; }
    }
    return best >= 0.70 ? best_tn : 0;
}

// 128 x 128 ring tile instead of 256 x 128 (bf16 C, both operands K-contiguous, a compile-time epilogue mode).
// Chosen where the 256 x 128 tiles would take at most 55 % of their 2 x #CU slots (the ping-pong tile has already been
// ruled out by then): tools/ring128_sweep.py, profiles/r02_ring128_sweep.txt -- 4096 x 768 x 3072: 60.6 -> 40.6 us,
// 8192 x 768 x 3072: 61.5 -> 49.5 us, 4096 x 1024 x 4096: 77.0 -> 53.6 us, 8192 x 1024 x 4096 (50 %): 81.0 -> 76.3 us.
static bool use_ring128(int M, int N, int K, int mode) {
    const int sel = polus_cfg().gemm_ring128;
    if (sel < 0 || mode < 0 || M < 128 || N < 128) return false;
    if (sel > 0) return true;
    const long slots = 2L * polus_num_cus();
    const long t256 = (long)((M + 255) / 256) * ((N + 127) / 128);
    return 20 * t256 <= 11 * slots;
}

// Number of K slices a bf16 Dense GEMM (both operands K-contiguous, bf16 C, any epilogue) should be cut into (f32 slabs
// + splitk_reduce_epi_kernel).  1 unless even the 128 x 128 ring tiles would take an eighth or less of their 3 x #CU
// slots (about two thousand tokens at N = 768) and K >= 2048; then enough slices to bring the 256 x 128 slab launch to
// ~3/4 of its slots, each at least 768 deep (<= 8).  Slicing wins in isolation over a wider range
// (profiles/r02_split_sweep.txt) but in the step its f32 slabs cost far more (profiles/r02_ab_auto_split.txt: 6144 and
// 8192 tokens lose 2.5-3 %), and the 128 x 128 tile beats it wherever that tile fills more than an eighth of the chip
// (profiles/r02_ring128_sweep.txt).  The caller passes the result as split_k with polus_gemm_workspace_bytes(M, N, split_k).
extern "C" int polus_gemm_auto_split(int M, int N, int K) {
    if (!polus_cfg().gemm_auto_split || M < 256 || N < 128 || K % 64 != 0 || K < 2048) return 1;
    if (pp_tile(M, N, K, 0, true)) return 1;
    const long ncu = polus_num_cus();
    if (polus_cfg().gemm_ring128 >= 0) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        if (8 * t128 > 3 * ncu) return 1;
    }
    const long slots = 2 * ncu;
    const long t = (long)((M + 255) / 256) * ((N + 127) / 128);
    if (4 * t > slots) return 1;
    long s = (slots * 3 / 4) / t;
    if (s > K / 768) s = K / 768;
    if (s > 8) s = 8;
    return s < 2 ? 1 : (int)s;
}

static int gemm_impl(int dtype, int a_layout, int b_layout, int c_dtype,
                     const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                     int M, int N, int K, float alpha,
                     const float* bias, const void* resid, long ldr, void* aux, long ldaux,
                     int act, int flags, int split_k, void* workspace, size_t workspace_bytes,
                     float drop_p, uint32_t seed, void* stream);

extern "C" int polus_gemm(int dtype, int a_layout, int b_layout, int c_dtype,
                          const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                          int M, int N, int K, float alpha,
                          const float* bias, const void* resid, long ldr, void* aux, long ldaux,
                          int act, int flags, int split_k, void* workspace, size_t workspace_bytes,
                          void* stream) {
    POLUS_REQUIRE(!(flags & POLUS_GEMM_DROPOUT), "polus_gemm: use polus_gemm_dropout for POLUS_GEMM_DROPOUT");
    return gemm_impl(dtype, a_layout, b_layout, c_dtype, A, lda, B, ldb, C, ldc, M, N, K, alpha, bias, resid, ldr,
                     aux, ldaux, act, flags, split_k, workspace, workspace_bytes, 0.0f, 0u, stream);
}

extern "C" int polus_gemm_dropout(int dtype, int a_layout, int b_layout, int c_dtype,
                                  const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                  int M, int N, int K, float alpha,
                                  const float* bias, const void* resid, long ldr, void* aux, long ldaux,
                                  int act, int flags, int split_k, void* workspace, size_t workspace_bytes,
                                  float drop_p, uint32_t seed, void* stream) {
    POLUS_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "polus_gemm_dropout: need 0 <= p < 1 (got %f)", drop_p);
    POLUS_REQUIRE((long)M * N < (1LL << 32), "polus_gemm_dropout: M*N must fit 32 bits");
    if (drop_p > 0.0f) flags |= POLUS_GEMM_DROPOUT; else flags &= ~POLUS_GEMM_DROPOUT;
    return gemm_impl(dtype, a_layout, b_layout, c_dtype, A, lda, B, ldb, C, ldc, M, N, K, alpha, bias, resid, ldr,
                     aux, ldaux, act, flags, split_k, workspace, workspace_bytes, drop_p, seed, stream);
}

static int gemm_impl(int dtype, int a_layout, int b_layout, int c_dtype,
                     const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                     int M, int N, int K, float alpha,
                     const float* bias, const void* resid, long ldr, void* aux, long ldaux,
                     int act, int flags, int split_k, void* workspace, size_t workspace_bytes,
                     float drop_p, uint32_t seed, void* stream) {
    POLUS_REQUIRE(dtype == POLUS_F32 || dtype == POLUS_BF16, "polus_gemm: bad dtype %d", dtype);
    POLUS_REQUIRE(c_dtype == POLUS_F32 || c_dtype == dtype, "polus_gemm: c_dtype must be f32 or dtype");
    POLUS_REQUIRE(M > 0 && N > 0 && K > 0, "polus_gemm: empty problem M=%d N=%d K=%d", M, N, K);
    POLUS_REQUIRE(A && B && C, "polus_gemm: null operand");
    POLUS_REQUIRE((a_layout | 1) == 1 && (b_layout | 1) == 1, "polus_gemm: bad layout");
    POLUS_REQUIRE(lda >= (a_layout == POLUS_K_CONTIG ? K : M), "polus_gemm: lda %ld too small", lda);
    POLUS_REQUIRE(ldb >= (b_layout == POLUS_K_CONTIG ? K : N), "polus_gemm: ldb %ld too small", ldb);
    POLUS_REQUIRE(ldc >= N, "polus_gemm: ldc %ld < N %d", ldc, N);
    POLUS_REQUIRE(!resid || ldr >= N, "polus_gemm: ldr too small");
    POLUS_REQUIRE(!((flags & POLUS_GEMM_ACT_BWD) && !aux), "polus_gemm: ACT_BWD needs aux");
    POLUS_REQUIRE(!aux || ldaux >= N, "polus_gemm: ldaux too small");
    if (split_k < 1) split_k = 1;
    const size_t es = polus_dtype_size(dtype), ecs = polus_dtype_size(c_dtype);
    const int bk = dtype == POLUS_BF16 ? 64 : 32;
    int nkt = (K + bk - 1) / bk;
    if (split_k > nkt) split_k = nkt;
    if (split_k > 1) {
        const bool epi = resid || aux || (flags & (POLUS_GEMM_ACT_FWD | POLUS_GEMM_ACT_BWD | POLUS_GEMM_DROPOUT));
        POLUS_REQUIRE(!epi || (dtype == POLUS_BF16 && c_dtype == POLUS_BF16 && a_layout == POLUS_K_CONTIG && b_layout == POLUS_K_CONTIG &&
                               !(flags & POLUS_GEMM_ACCUM_C)),
                      "polus_gemm: split_k with a residual / activation / dropout epilogue needs bf16 K-contiguous operands and a bf16 C");
        if (!workspace || workspace_bytes < polus_gemm_workspace_bytes(M, N, split_k)) {
            polus_set_error("polus_gemm: workspace %zu < %zu", workspace_bytes, polus_gemm_workspace_bytes(M, N, split_k));
            return POLUS_ERR_WORKSPACE;
        }
    }
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.alpha = alpha;
    a.k_per_split = ((nkt + split_k - 1) / split_k) * bk;
    a.bias = bias; a.resid = resid; a.ldr = ldr; a.aux = aux; a.ldaux = ldaux;
    a.act = act; a.flags = flags;
    a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_thresh = polus_drop_thresh(drop_p); a.drop_seed = seed;
    a.dyn = polus_dyn();
    a.partial = split_k > 1 ? static_cast<float*>(workspace) : nullptr;
    const int epc = (int)(16 / es);
    // whole-chunk validity: the contiguous extent of each operand must be a multiple of a chunk
    a.a_vec = polus_aligned16(A) && ((lda * es) % 16 == 0) && ((a_layout == POLUS_K_CONTIG ? K : M) % epc == 0);
    a.b_vec = polus_aligned16(B) && ((ldb * es) % 16 == 0) && ((b_layout == POLUS_K_CONTIG ? K : N) % epc == 0);
    // 4-wide epilogue accesses: every touched row start must be 4-element aligned
    bool ev = (((uintptr_t)C) % (4 * ecs) == 0) && (ldc % 4 == 0);
    if (bias) ev = ev && (((uintptr_t)bias) % 16 == 0);
    if (resid) ev = ev && (((uintptr_t)resid) % (4 * es) == 0) && (ldr % 4 == 0);
    if (aux) ev = ev && (((uintptr_t)aux) % (4 * es) == 0) && (ldaux % 4 == 0);
    a.epi_vec = ev;
    bool ev16 = (((uintptr_t)C) % 16 == 0) && ((ldc * ecs) % 16 == 0);
    if (bias) ev16 = ev16 && (((uintptr_t)bias) % 16 == 0);
    if (resid) ev16 = ev16 && (((uintptr_t)resid) % 16 == 0) && ((ldr * es) % 16 == 0);
    if (aux) ev16 = ev16 && (((uintptr_t)aux) % 16 == 0) && ((ldaux * es) % 16 == 0);
    a.epi_vec16 = ev16 && dtype == POLUS_BF16;
    a.order = polus_cfg().gemm_order;
    a.persist = polus_cfg().gemm_persist ? max(32, polus_num_cus() - polus_reserved_cus()) : 0;
    a.persist_all = polus_cfg().gemm_persist >= 2;
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int splits_eff = (nkt + (a.k_per_split / bk) - 1) / (a.k_per_split / bk);
    dim3 grid(tiles, 1, split_k > 1 ? splits_eff : 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = POLUS_OK;
    a.c_split_stride = 0;
    a.colsum_a = nullptr;
    const bool both_kc = a_layout == POLUS_K_CONTIG && b_layout == POLUS_K_CONTIG;
    const bool fast_bf16 = dtype == POLUS_BF16 && a.a_vec && a.b_vec && M >= 256 && N >= 128 && !polus_cfg().gemm_v1;
    const bool epi_split = split_k > 1 && (resid || aux || (flags & (POLUS_GEMM_ACT_FWD | POLUS_GEMM_ACT_BWD | POLUS_GEMM_DROPOUT)));
    if (epi_split && fast_bf16 && both_kc && c_dtype == POLUS_BF16) {
        GemmArgs s = a;              // slabs: plain f32 stores of every K slice ...
        s.C = a.partial; s.ldc = N; s.c_split_stride = (long)M * N; s.partial = nullptr;
        s.alpha = 1.0f; s.bias = nullptr; s.flags = 0; s.resid = nullptr; s.aux = nullptr;
        s.epi_vec = (N % 4 == 0); s.epi_vec16 = (N % 4 == 0) && polus_aligned16(a.partial);
        rc = polus_launch_gemm_ring(s, 1, 0, 0, splits_eff, st);
        if (rc != POLUS_OK) return rc;
        GemmArgs e = a;              // ... then the whole epilogue on their sum
        e.partial = nullptr;
        const long quads = (long)M * ((N + 3) / 4);
        const int blocks = (int)((quads + 255) / 256);
        if (flags & POLUS_GEMM_DROPOUT)
            hipLaunchKernelGGL(splitk_reduce_epi_kernel<true>, dim3(blocks), dim3(256), 0, st, e, a.partial, splits_eff);
        else
            hipLaunchKernelGGL(splitk_reduce_epi_kernel<false>, dim3(blocks), dim3(256), 0, st, e, a.partial, splits_eff);
        POLUS_CHECK_LAUNCH("polus_gemm(splitk_reduce_epi)");
        return POLUS_OK;
    }
    if (epi_split) {                 // not on the fast path (small or unaligned problem): one slice, in-kernel epilogue
        split_k = 1; a.partial = nullptr;
        a.k_per_split = nkt * bk;
        grid = dim3(tiles, 1, 1);
        splits_eff = 1;
    }
    if (flags & POLUS_GEMM_DROPOUT) {
        // forward Dense only: K-contiguous operands, C in the compute dtype
        POLUS_REQUIRE(both_kc && c_dtype == dtype, "polus_gemm_dropout: needs K-contiguous operands and c_dtype == dtype");
        if (dtype == POLUS_BF16 && a.a_vec && a.b_vec && M >= 256 && N >= 128 && !polus_cfg().gemm_v1) {
            a.k_per_split = ((K + 31) / 32) * 32;
            if (const int tn = pp_tile(M, N, K, polus_gemm_epi_mode(a, 0, 1), a.epi_vec16))
                return polus_launch_gemm_pp(a, polus_gemm_epi_mode(a, 0, 1), 1, tn, st);
            if (use_ring128(M, N, K, polus_gemm_epi_mode(a, 0, 1))) return polus_launch_gemm_ring128(a, polus_gemm_epi_mode(a, 0, 1), 1, st);
            return polus_launch_gemm_ring_dropout(a, st);
        }
        const bool v = a.a_vec && a.b_vec;
        if (dtype == POLUS_BF16) return v ? launch_v<bf16_t, false, false, bf16_t, true, true>(a, grid, st) : launch_v<bf16_t, false, false, bf16_t, false, true>(a, grid, st);
        return v ? launch_v<float, false, false, float, true, true>(a, grid, st) : launch_v<float, false, false, float, false, true>(a, grid, st);
    }
    const int a_ks = a_layout == POLUS_K_STRIDED, b_ks = b_layout == POLUS_K_STRIDED;
    if (dtype == POLUS_BF16 && a.a_vec && a.b_vec && M >= 256 && N >= 128 && !polus_cfg().gemm_v1) {
        if (split_k <= 1) {
            a.k_per_split = ((K + 31) / 32) * 32;
            if (both_kc) {
                if (const int tn = pp_tile(M, N, K, polus_gemm_epi_mode(a, c_dtype == POLUS_F32, 0), a.epi_vec16))
                    return polus_launch_gemm_pp(a, polus_gemm_epi_mode(a, c_dtype == POLUS_F32, 0), 0, tn, st);
            }
            if (both_kc && c_dtype == POLUS_BF16 && use_ring128(M, N, K, polus_gemm_epi_mode(a, 0, 0)))
                return polus_launch_gemm_ring128(a, polus_gemm_epi_mode(a, 0, 0), 0, st);
            return polus_launch_gemm_ring(a, c_dtype == POLUS_F32, a_ks, b_ks, 1, st);
        }
        GemmArgs s = a;              // slabs: plain f32 stores, epilogue applied by the reduce kernel
        s.C = a.partial; s.ldc = N; s.c_split_stride = (long)M * N; s.partial = nullptr;
        s.alpha = 1.0f; s.bias = nullptr; s.flags = 0; s.resid = nullptr; s.aux = nullptr;
        s.epi_vec = (N % 4 == 0); s.epi_vec16 = (N % 4 == 0) && polus_aligned16(a.partial);
        rc = polus_launch_gemm_ring(s, 1, a_ks, b_ks, splits_eff, st);
        if (rc != POLUS_OK) return rc;
        goto reduce;
    }
    if (dtype == POLUS_BF16) {
        rc = (c_dtype == POLUS_F32) ? dispatch_layout<bf16_t, float>(a_layout, b_layout, a, grid, st)
                                    : dispatch_layout<bf16_t, bf16_t>(a_layout, b_layout, a, grid, st);
    } else {
        rc = dispatch_layout<float, float>(a_layout, b_layout, a, grid, st);
    }
    if (rc != POLUS_OK) return rc;
reduce:
    if (split_k > 1) {
        long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (c_dtype == POLUS_F32)
            hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st,
                               a.partial, splits_eff, M, N, static_cast<float*>(C), ldc, alpha, bias,
                               (flags & POLUS_GEMM_ACCUM_C) ? 1 : 0);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st,
                               a.partial, splits_eff, M, N, static_cast<bf16_t*>(C), ldc, alpha, bias,
                               (flags & POLUS_GEMM_ACCUM_C) ? 1 : 0);
        POLUS_CHECK_LAUNCH("polus_gemm(splitk_reduce)");
    }
    return POLUS_OK;
}

// ---- Dense backward for the parameters: dW = dY^T X (+)= and db = column sums of dY, one pass
// over dY.  On the ring kernel the bias gradient rides on the matrix pipe (ones-fragment MFMA).
extern "C" size_t polus_dense_bwd_params_workspace_bytes(int T, int n_out, int n_in, int split_k) {
    if (split_k < 1) split_k = 1;
    size_t slabs = split_k > 1 ? (size_t)split_k * n_out * n_in * sizeof(float) : 0;
    size_t cs = (size_t)split_k * n_out * sizeof(float);
    size_t fallback = polus_colsum_workspace_bytes(T, n_out);
    return slabs + (cs > fallback ? cs : fallback) + 256;
}

namespace {
__global__ void colsum_splits_kernel(const float* __restrict__ partial, int splits, int n, float* __restrict__ out, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(long)z * n + c];
    out[c] = accumulate ? out[c] + s : s;
}
}  // namespace

extern "C" int polus_dense_bwd_params(int dtype, const void* dY, long lddy, const void* X, long ldx,
                                      float* dW, long lddw, float* db, int T, int n_out, int n_in,
                                      int accumulate, int split_k, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    POLUS_REQUIRE(dY && X && dW, "polus_dense_bwd_params: null pointer");
    POLUS_REQUIRE(T > 0 && n_out > 0 && n_in > 0, "polus_dense_bwd_params: bad shape");
    if (split_k < 1) split_k = 1;
    size_t need = polus_dense_bwd_params_workspace_bytes(T, n_out, n_in, split_k);
    if (!workspace || workspace_bytes < need) { polus_set_error("polus_dense_bwd_params: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t es = polus_dtype_size(dtype);
    const int bk = dtype == POLUS_BF16 ? 64 : 32;
    int nkt = (T + bk - 1) / bk;
    if (split_k > nkt) split_k = nkt;
    const size_t slab_bytes = split_k > 1 ? (((size_t)split_k * n_out * n_in * sizeof(float) + 255) / 256) * 256 : 0;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    float* cs_ws = reinterpret_cast<float*>(ws + slab_bytes);
    const bool vec = polus_aligned16(dY) && polus_aligned16(X) && ((lddy * es) % 16 == 0) && ((ldx * es) % 16 == 0) &&
                     (n_out % (16 / es) == 0) && (n_in % (16 / es) == 0);
    const bool ring = dtype == POLUS_BF16 && vec && n_out >= 256 && n_in >= 128 && db != nullptr && !polus_cfg().gemm_v1;
    if (!ring) {
        int rc = polus_gemm(dtype, POLUS_K_STRIDED, POLUS_K_STRIDED, POLUS_F32, dY, lddy, X, ldx, dW, lddw, n_out, n_in, T,
                            1.0f, nullptr, nullptr, 0, nullptr, 0, 0, accumulate ? POLUS_GEMM_ACCUM_C : 0, split_k,
                            ws, slab_bytes, stream);
        if (rc != POLUS_OK || !db) return rc;
        return polus_colsum(dtype, dY, lddy, T, n_out, db, accumulate, cs_ws, workspace_bytes - slab_bytes, stream);
    }
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = dY; a.B = X; a.lda = lddy; a.ldb = ldx;
    a.M = n_out; a.N = n_in; a.K = T; a.alpha = 1.0f;
    a.k_per_split = ((nkt + split_k - 1) / split_k) * bk;
    const int splits_eff = (nkt + (a.k_per_split / bk) - 1) / (a.k_per_split / bk);
    a.a_vec = a.b_vec = 1;
    a.colsum_a = cs_ws;
    a.order = polus_cfg().gemm_order;
    a.persist = polus_cfg().gemm_persist ? max(32, polus_num_cus() - polus_reserved_cus()) : 0;
    a.persist_all = polus_cfg().gemm_persist >= 2;
    int rc;
    if (splits_eff > 1) {
        a.C = ws; a.ldc = n_in; a.c_split_stride = (long)n_out * n_in;
        a.epi_vec = (n_in % 4 == 0); a.epi_vec16 = (n_in % 4 == 0);
        rc = polus_launch_gemm_ring(a, 1, 1, 1, splits_eff, st);
        if (rc != POLUS_OK) return rc;
        long total = (long)n_out * n_in;
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((int)((total + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const float*>(ws), splits_eff, n_out, n_in, dW, lddw, 1.0f,
                           (const float*)nullptr, accumulate ? 1 : 0);
        POLUS_CHECK_LAUNCH("polus_dense_bwd_params(reduce)");
    } else {
        a.C = dW; a.ldc = lddw; a.flags = accumulate ? POLUS_GEMM_ACCUM_C : 0;
        a.epi_vec = polus_aligned16(dW) && (lddw % 4 == 0); a.epi_vec16 = a.epi_vec;
        rc = polus_launch_gemm_ring(a, 1, 1, 1, 1, st);
        if (rc != POLUS_OK) return rc;
    }
    hipLaunchKernelGGL(colsum_splits_kernel, dim3((n_out + 255) / 256), dim3(256), 0, st, cs_ws, splits_eff, n_out, db, accumulate ? 1 : 0);
    POLUS_CHECK_LAUNCH("polus_dense_bwd_params(colsum)");
    return POLUS_OK;
}

// ---- The same for several Dense layers at once (all four of an encoder layer): one ring launch
// over the concatenated (split, tile) lists, so the chip fills with 2-3 K-splits instead of 7-28 per
// matrix (each split is an f32 slab written and read back).  All problems share T and accumulate.
// split_k > 0: that many splits for every problem; split_k <= 0: chosen per problem so that the
// launch is one full round of the 2 x #CU workgroup slots.
// Which kernel runs the grouped dW launch: the ping-pong 256 x 256 kernel (gemm_ppks.hip, one workgroup per CU)
// when every problem is at least one tile in both directions and T is whole K-tiles, else the ring kernel
// (256 x 128 tiles, two workgroups per CU).
static bool grouped_use_pp(int n, const polus_dw_problem* pr, int T) {
    if (polus_cfg().gemm_pp < 0 || T % 64 != 0) return false;
    for (int k = 0; k < n; ++k)
        if (pr[k].n_out < 256 || pr[k].n_in < 256) return false;
    return true;
}

// The stream-K hybrid replaces the even split when the library chooses the slices itself (split_k <= 0), the ping-pong
// kernel applies and the even split would leave at least 1/16 of the CUs without a workgroup.
static bool grouped_sk_plan(int n, const polus_dw_problem* pr, int T, int split_k, PPKSSKPlan* pl) {
    if (!polus_cfg().dw_streamk || split_k > 0 || !grouped_use_pp(n, pr, T)) return false;
    int no[POLUS_MAX_GROUP], ni[POLUS_MAX_GROUP];
    for (int k = 0; k < n; ++k) { no[k] = pr[k].n_out; ni[k] = pr[k].n_in; }
    int ncu = polus_num_cus() - polus_reserved_cus();
    if (polus_cfg().dw_sk_cus > 0 && polus_cfg().dw_sk_cus < ncu) ncu = polus_cfg().dw_sk_cus;
    return polus_ppks_sk_plan(no, ni, n, T, ncu, polus_cfg().dw_sk_delta, pl) != 0;
}
static size_t grouped_sk_need(int n, const polus_dw_problem* pr, const PPKSSKPlan& pl, size_t* cs_off) {
    size_t off = (size_t)pl.ttot * pl.slots * 256 * 256 * sizeof(float);
    for (int k = 0; k < n; ++k) {
        cs_off[k] = off;
        off += (((size_t)pl.slots * pr[k].n_out * sizeof(float) + 255) / 256) * 256;
    }
    return off + 256;
}

static void grouped_splits(int n, const polus_dw_problem* pr, int T, int split_k, int* splits, bool pp) {
    const int nkt = (T + 63) / 64;
    int tiles[POLUS_MAX_GROUP], total = 0;
    for (int k = 0; k < n; ++k) {
        tiles[k] = pp ? polus_ppks_tiles(pr[k].n_out, pr[k].n_in) : ((pr[k].n_out + 255) / 256) * ((pr[k].n_in + 127) / 128);
        total += tiles[k];
    }
    if (split_k > 0) {
        for (int k = 0; k < n; ++k) splits[k] = split_k < nkt ? split_k : nkt;
        return;
    }
    // Workgroups go to the 8 XCDs round-robin and each tile list is padded to a multiple of 8, so
    // XCD 0 receives ceil(tiles/8) workgroups of every (problem, split).  All workgroups run for
    // about the same (long) time: one more than the resident workgroups per XCD (2 per CU for the ring
    // kernel, 1 for the ping-pong kernel) on any XCD doubles the kernel.
    const int slots_xcd = (pp ? 1 : 2) * polus_num_cus() / 8;
    int per_xcd[POLUS_MAX_GROUP], total_xcd = 0;
    for (int k = 0; k < n; ++k) { per_xcd[k] = (tiles[k] + 7) / 8; total_xcd += per_xcd[k]; }
    int base = slots_xcd / (total_xcd > 0 ? total_xcd : 1);
    if (base < 1) base = 1;
    const int cap = nkt / 4 > 0 ? nkt / 4 : 1;          // >= 256 contraction rows per split
    if (base > cap) base = cap;
    int used = 0;
    for (int k = 0; k < n; ++k) { splits[k] = base; used += per_xcd[k] * base; }
    // hand the remaining slots to the problems with the most tiles that still fit, one extra split each
    bool given[POLUS_MAX_GROUP] = {false};
    for (;;) {
        int best = -1;
        for (int k = 0; k < n; ++k)
            if (!given[k] && splits[k] < cap && used + per_xcd[k] <= slots_xcd && (best < 0 || tiles[k] > tiles[best])) best = k;
        if (best < 0) break;
        given[best] = true; ++splits[best]; used += per_xcd[best];
    }
}

static size_t grouped_need(int n, const polus_dw_problem* pr, const int* splits, size_t* slab_off, size_t* cs_off) {
    size_t off = 0;
    for (int k = 0; k < n; ++k) {
        slab_off[k] = off;
        if (splits[k] > 1) off += (((size_t)splits[k] * pr[k].n_out * pr[k].n_in * sizeof(float) + 255) / 256) * 256;
        cs_off[k] = off;
        off += (((size_t)splits[k] * pr[k].n_out * sizeof(float) + 255) / 256) * 256;
    }
    return off + 256;
}

extern "C" size_t polus_dense_bwd_params_grouped_workspace_bytes(int n, const polus_dw_problem* problems, int T, int split_k) {
    if (n < 1 || n > POLUS_MAX_GROUP || !problems || T < 1) return 0;
    int splits[POLUS_MAX_GROUP];
    grouped_splits(n, problems, T, split_k, splits, grouped_use_pp(n, problems, T));
    size_t so[POLUS_MAX_GROUP], co[POLUS_MAX_GROUP];
    size_t grouped = grouped_need(n, problems, splits, so, co);
    size_t single = 0;   // fallback path runs them one by one
    for (int k = 0; k < n; ++k) {
        size_t b = polus_dense_bwd_params_workspace_bytes(T, problems[k].n_out, problems[k].n_in, splits[k]);
        if (b > single) single = b;
    }
    PPKSSKPlan pl;
    if (grouped_sk_plan(n, problems, T, split_k, &pl)) {
        size_t co2[POLUS_MAX_GROUP];
        const size_t sk = grouped_sk_need(n, problems, pl, co2);
        if (sk > grouped) grouped = sk;
    }
    return grouped > single ? grouped : single;
}

extern "C" int polus_dense_bwd_params_grouped(int dtype, int n, const polus_dw_problem* problems, int T, int accumulate,
                                              int split_k, void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(problems && n >= 1 && n <= POLUS_MAX_GROUP && T > 0, "polus_dense_bwd_params_grouped: bad arguments");
    const size_t es = polus_dtype_size(dtype);
    bool ring = dtype == POLUS_BF16 && !polus_cfg().gemm_v1 && !polus_cfg().dw_ungrouped;
    for (int k = 0; k < n; ++k) {
        const polus_dw_problem& q = problems[k];
        POLUS_REQUIRE(q.dY && q.X && q.dW && q.n_out > 0 && q.n_in > 0, "polus_dense_bwd_params_grouped: problem %d: bad arguments", k);
        ring = ring && polus_aligned16(q.dY) && polus_aligned16(q.X) && ((q.lddy * es) % 16 == 0) && ((q.ldx * es) % 16 == 0) &&
               (q.n_out % 8 == 0) && (q.n_in % 8 == 0) && q.n_out >= 256 && q.n_in >= 128;
    }
    const bool pp = ring && grouped_use_pp(n, problems, T);
    int splits[POLUS_MAX_GROUP];
    grouped_splits(n, problems, T, split_k, splits, grouped_use_pp(n, problems, T));
    size_t need = polus_dense_bwd_params_grouped_workspace_bytes(n, problems, T, split_k);
    if (!workspace || workspace_bytes < need) { polus_set_error("polus_dense_bwd_params_grouped: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    if (!ring) {
        for (int k = 0; k < n; ++k) {
            const polus_dw_problem& q = problems[k];
            int rc;
            if (q.db) rc = polus_dense_bwd_params(dtype, q.dY, q.lddy, q.X, q.ldx, q.dW, q.lddw, q.db, T, q.n_out, q.n_in,
                                                  accumulate, splits[k], workspace, workspace_bytes, stream);
            else rc = polus_gemm(dtype, POLUS_K_STRIDED, POLUS_K_STRIDED, POLUS_F32, q.dY, q.lddy, q.X, q.ldx, q.dW, q.lddw,
                                 q.n_out, q.n_in, T, 1.0f, nullptr, nullptr, 0, nullptr, 0, 0, accumulate ? POLUS_GEMM_ACCUM_C : 0,
                                 splits[k], workspace, workspace_bytes, stream);
            if (rc != POLUS_OK) return rc;
        }
        return POLUS_OK;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    PPKSSKPlan skp;
    if (pp && grouped_sk_plan(n, problems, T, split_k, &skp)) {
        size_t co2[POLUS_MAX_GROUP];
        grouped_sk_need(n, problems, skp, co2);
        unsigned char* ws2 = static_cast<unsigned char*>(workspace);
        GemmArgs ga2[POLUS_MAX_GROUP];
        float* cs2[POLUS_MAX_GROUP]; float* dWs[POLUS_MAX_GROUP]; float* dbs[POLUS_MAX_GROUP];
        long ldw[POLUS_MAX_GROUP]; int no[POLUS_MAX_GROUP], ni[POLUS_MAX_GROUP];
        bool aligned = true;
        for (int k = 0; k < n; ++k) {
            const polus_dw_problem& q = problems[k];
            GemmArgs& a = ga2[k];
            memset(&a, 0, sizeof(a));
            a.A = q.dY; a.B = q.X; a.lda = q.lddy; a.ldb = q.ldx;
            a.M = q.n_out; a.N = q.n_in; a.K = T; a.alpha = 1.0f;
            a.a_vec = a.b_vec = 1; a.epi_vec = a.epi_vec16 = 1;
            cs2[k] = q.db ? reinterpret_cast<float*>(ws2 + co2[k]) : nullptr;
            dWs[k] = q.dW; dbs[k] = q.db; ldw[k] = q.lddw; no[k] = q.n_out; ni[k] = q.n_in;
            aligned = aligned && (q.n_in % 4 == 0) && (q.lddw % 4 == 0) && polus_aligned16(q.dW);
        }
        if (aligned) {
            int rc2 = polus_launch_gemm_ppks_sk(ga2, skp, reinterpret_cast<float*>(ws2), cs2, st);
            if (rc2 != POLUS_OK) return rc2;
            return polus_launch_dw_group_reduce_sk(skp, reinterpret_cast<const float*>(ws2), cs2, dWs, dbs, ldw, no, ni, accumulate ? 1 : 0, st);
        }
    }
    const int bk = 64;
    const int nkt = (T + bk - 1) / bk;
    int eff[POLUS_MAX_GROUP], kps[POLUS_MAX_GROUP];
    for (int k = 0; k < n; ++k) {
        int sk = splits[k] > nkt ? nkt : splits[k];
        kps[k] = ((nkt + sk - 1) / sk) * bk;
        eff[k] = (nkt + (kps[k] / bk) - 1) / (kps[k] / bk);
    }
    size_t so[POLUS_MAX_GROUP], co[POLUS_MAX_GROUP];
    grouped_need(n, problems, splits, so, co);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    GemmArgs ga[POLUS_MAX_GROUP];
    for (int k = 0; k < n; ++k) {
        const polus_dw_problem& q = problems[k];
        GemmArgs& a = ga[k];
        memset(&a, 0, sizeof(a));
        a.A = q.dY; a.B = q.X; a.lda = q.lddy; a.ldb = q.ldx;
        a.M = q.n_out; a.N = q.n_in; a.K = T; a.alpha = 1.0f;
        a.k_per_split = kps[k];
        a.a_vec = a.b_vec = 1;
        a.colsum_a = q.db ? reinterpret_cast<float*>(ws + co[k]) : nullptr;
        a.order = 0;
        a.persist = 0;
        a.persist_all = 0;
        if (eff[k] > 1) {
            a.C = ws + so[k]; a.ldc = q.n_in; a.c_split_stride = (long)q.n_out * q.n_in;
            a.epi_vec = a.epi_vec16 = (q.n_in % 4 == 0);
        } else {
            a.C = q.dW; a.ldc = q.lddw; a.flags = accumulate ? POLUS_GEMM_ACCUM_C : 0;
            a.epi_vec = polus_aligned16(q.dW) && (q.lddw % 4 == 0); a.epi_vec16 = a.epi_vec;
        }
    }
    int rc = pp ? polus_launch_gemm_ppks_grouped_dw(ga, n, eff, st) : polus_launch_gemm_ring_grouped_dw(ga, n, eff, st);
    if (rc != POLUS_OK) return rc;
    if (polus_cfg().dw_fused_reduce) {
        // every slab reduction and every bias-gradient finalisation of the group in ONE launch
        const float* slabs[POLUS_MAX_GROUP]; const float* cs[POLUS_MAX_GROUP];
        float* dWs[POLUS_MAX_GROUP]; float* dbs[POLUS_MAX_GROUP];
        long ldw[POLUS_MAX_GROUP]; int no[POLUS_MAX_GROUP], ni[POLUS_MAX_GROUP];
        bool fusable = true;
        for (int k = 0; k < n; ++k) {
            const polus_dw_problem& q = problems[k];
            slabs[k] = eff[k] > 1 ? reinterpret_cast<const float*>(ws + so[k]) : nullptr;
            cs[k] = q.db ? reinterpret_cast<const float*>(ws + co[k]) : nullptr;
            dWs[k] = q.dW; dbs[k] = q.db; ldw[k] = q.lddw; no[k] = q.n_out; ni[k] = q.n_in;
            fusable = fusable && (q.n_in % 4 == 0) && (q.lddw % 4 == 0) && polus_aligned16(q.dW);
        }
        if (fusable) return polus_launch_dw_group_reduce(n, slabs, cs, dWs, dbs, ldw, no, ni, eff, accumulate ? 1 : 0, st);
    }
    for (int k = 0; k < n; ++k) {
        const polus_dw_problem& q = problems[k];
        if (eff[k] > 1) {
            long total = (long)q.n_out * q.n_in;
            hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((int)((total + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const float*>(ws + so[k]), eff[k], q.n_out, q.n_in, q.dW, q.lddw, 1.0f,
                               (const float*)nullptr, accumulate ? 1 : 0);
            POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(reduce)");
        }
        if (q.db) {
            hipLaunchKernelGGL(colsum_splits_kernel, dim3((q.n_out + 255) / 256), dim3(256), 0, st,
                               reinterpret_cast<const float*>(ws + co[k]), eff[k], q.n_out, q.db, accumulate ? 1 : 0);
            POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(colsum)");
        }
    }
    return POLUS_OK;
}
