// Weight gradients dW = dY^T X of several Dense layers in one launch, 256 x 256 tile, 8 waves in two
// half-phase-staggered groups (gfx950) -- the K-strided sibling of gemm_pp.hip.
//
// Both operands are K-strided: A = dY stored [T][n_out], B = X stored [T][n_in], contraction over T
// (16384 rows at the headline shape: 256 K-tiles of 64).  Their LDS images stay k-major (a K-tile of an
// operand = 64 k-rows x 512 B = 32 KiB; an LDS-DMA wave-instruction = 2 whole k-rows = full 128-byte lines
// from HBM) and fragments come out of ds_read_b64_tr_b16, so no transposed copy of an activation ever
// exists.  The 32-byte unit index inside a k-row is XORed with (k & 3) | ((k >> 3) & 1) << 2 on the DMA
// source side and on the read side: the 8 k-rows a half-wave reads fall on 8 distinct bank windows.
//
// Waves 2(M) x 4(N), 128 x 64 per wave (8 x 4 MFMA 16x16x32 tiles, 128 accumulator registers), groups
// = wm as in gemm_pp.hip.  A phase is one K-HALF of a K-tile: 24 transposing reads (8 A + 4 B fragments),
// 32 MFMAs (+2 for the bias gradient, below).  Phase j consumes k-rows 32 (j & 1).. of stage (j >> 1) & 1
// of both operands; that 2 x 16 KiB region is refilled one phase later (4 LDS-DMA per wave) with the data
// of phase j + 4, so every byte has three phases (~3k cycles) to land and the only wait is `vmcnt(8)` at the
// end of each R section.  R sections run at raised priority (see gemm_pp.hip).
//
// Bias gradient: db[m] = sum_t dY[t][m] rides on the matrix pipe: in tiles of the first tile column, wave
// (wm, wn) adds one MFMA against a ones-fragment for m-tiles 2wn and 2wn + 1 of its half per phase.
//
// Output: f32.  K is cut into `splits` slices per problem (chosen so that the launch fills the chip once);
// a slice writes an f32 slab (or, when a problem has one slice, dW itself), and ONE reduce launch for the
// whole group adds the slabs in slice order and finishes the bias gradients (dw_group_reduce_kernel).
#include <cstddef>
#include <type_traits>
#include "gemm_common.h"

using namespace pgemm;

namespace {

constexpr int TM = 256, TN = 256, TK = 64, NTHR = 512;
constexpr int HALF = 32 * 512;                // one k-half of one operand
constexpr int REGION = 2 * HALF;              // one operand, one stage
constexpr int STAGE = 2 * REGION;
constexpr int SMEM = 2 * STAGE;               // 128 KiB

__device__ const uint4 g_zero_chunk_ppks[1] = {{0u, 0u, 0u, 0u}};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int N> __device__ __forceinline__ void vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ void ppks_body(const GemmArgs& p, const int wg_in, const int nwg, const int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 2, wn = wid & 3;

    const int tiles_n = (p.N + TN - 1) / TN;
    const int wg = xcd_remap(wg_in, nwg);
    const int m0 = (wg / tiles_n) * TM, n0 = (wg % tiles_n) * TN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int NP = (kend - kbeg) / 32;                 // phases (K-halves); even: k ranges are whole K-tiles
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_chunk_ppks);

    // ---- LDS-DMA: a phase's half-region of an operand = 32 k-rows x 512 B = 2 wave-instructions per wave;
    // wave w, piece j covers k-rows 16 j + 2 w + (lane >> 5), LDS chunk lane & 31
    const bf16_t* src[4];          // [A0, A1, B0, B1] at phase 0
    bool ok[4];
    int dst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool isA = j < 2;
        const int kr = 16 * (j & 1) + 2 * wid + (lane >> 5);       // k-row within the half
        const int f = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int lc = (lane & 31) ^ (f << 1);
        const int col = (isA ? m0 : n0) + lc * 8;
        ok[j] = col < (isA ? p.M : p.N);
        src[j] = static_cast<const bf16_t*>(isA ? p.A : p.B) + (long)(kbeg + kr) * (isA ? p.lda : p.ldb) + col;
        dst[j] = (isA ? 0 : REGION) + (16 * (j & 1) + 2 * wid) * 512;
    }
    const long a_step = 32L * p.lda, b_step = 32L * p.ldb;       // one phase further along K
    auto issue = [&](int ph) {
        unsigned char* base = smem + ((ph >> 1) & 1) * STAGE + (ph & 1) * HALF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16_t* sp = ok[j] ? src[j] + (long)ph * (j < 2 ? a_step : b_step) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lds_void_t*)(base + dst[j]), 16, 0, 0);
        }
    };

    // ---- fragment addresses (transposing read): lane (i, g) passes k-row 8g + (i >> 2) (+4 for the second
    // read), columns 4 (i & 3)..+3 of the 16-column unit of its tile
    const int krow = 8 * g + (i >> 2);
    const int fx = ((krow & 3) | (((krow >> 3) & 1) << 2)) << 5;     // same for krow + 4
    const int a_col = (wm * 128) * 2 + (i & 3) * 8;                  // + mt * 32, then ^ fx
    const int b_col = (wn * 64) * 2 + (i & 3) * 8;                   // + nt * 32, then ^ fx

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool want_colsum = p.colsum_a != nullptr && n0 == 0;
    f32x4 csum[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    Frag<bf16_t> ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones.v[e] = (bf16_t)1.0f;
    Frag<bf16_t> af[8], bfr[4];

    // ---- prologue: phases 0..2 in flight, phase 0 landed
    issue(0);
    if (NP > 1) issue(1);
    if (NP > 2) { issue(2); vmcnt<8>(); } else if (NP > 1) { vmcnt<4>(); } else { vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier behind

    // REM = phases after this one that still have to be issued or landed: 0 steady (ph + 3 < NP),
    // 1: ph + 3 == NP, 2: ph + 2 == NP, 3: last phase
    auto phase = [&](auto REM_, int ph) {
        constexpr int REM = decltype(REM_)::value;
        const unsigned char* st = smem + ((ph >> 1) & 1) * STAGE + (ph & 1) * HALF;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const unsigned char* q = st + REGION + krow * 512 + ((b_col + nt * 32) ^ fx);
            s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * 512);
            s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bfr[nt].v = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const unsigned char* q = st + krow * 512 + ((a_col + mt * 32) ^ fx);
            s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * 512);
            s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            af[mt].v = __builtin_bit_cast(bf16x8, w);
        }
        if (REM == 0) issue(ph + 3);        // into the region the previous phase's reads released
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (REM == 0) vmcnt<8>(); else if (REM == 1) vmcnt<4>(); else if (REM == 2) vmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) mma16(acc[mt][nt], bfr[nt], af[mt]);
        if (want_colsum) {
            // wave (wm, wn) sums m-tiles 2wn and 2wn + 1 of its half: a register array indexed by wn would go to scratch
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (wn == w) { mma16(csum[0], ones, af[2 * w]); mma16(csum[1], ones, af[2 * w + 1]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    typedef std::integral_constant<int, 0> R0;
    int ph = 0;
    for (; ph + 3 < NP; ++ph) phase(R0{}, ph);
    if (ph + 3 == NP) { phase(std::integral_constant<int, 1>{}, ph); ++ph; }
    if (ph + 2 == NP) { phase(std::integral_constant<int, 2>{}, ph); ++ph; }
    phase(std::integral_constant<int, 3>{}, ph);

    if (wm == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's last barrier
    if (want_colsum && g == 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int m = m0 + wm * 128 + (2 * wn + e) * 16 + i;
            if (m < p.M) p.colsum_a[(long)split * p.M + m] = csum[e][0];
        }
    }
    GemmArgs q = p;
    q.C = static_cast<float*>(p.C) + (long)split * p.c_split_stride;
    epilogue_wave_128x64_lds<float, false>(q, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wid * 8704);
}

struct PPKSGroupArgs {
    GemmArgs p[POLUS_MAX_GROUP];
    int wg0[POLUS_MAX_GROUP + 1];     // first workgroup of each problem, ascending
    int tiles[POLUS_MAX_GROUP];       // real tiles of each problem
    int tpad[POLUS_MAX_GROUP];        // tiles rounded up to a multiple of 8 (workgroup index mod 8 stays the XCD)
    int n;
};

__global__ __launch_bounds__(NTHR, 2) void gemm_ppks_grouped_kernel(PPKSGroupArgs ga) {
    const int b = blockIdx.x;
    int q = 0;
#pragma unroll
    for (int k = 1; k < POLUS_MAX_GROUP; ++k)
        if (k < ga.n && b >= ga.wg0[k]) q = k;
    const int rel = b - ga.wg0[q];
    const int split = rel / ga.tpad[q], wg = rel - split * ga.tpad[q];
    if (wg >= ga.tiles[q]) return;          // padding workgroup
    // the chosen problem's arguments straight out of the kernarg segment (a run-time index into the by-value
    // array would go through scratch)
    typedef const __attribute__((address_space(4))) unsigned char* karg_t;
    karg_t ka = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    GemmArgs P;
    {
        static_assert(sizeof(GemmArgs) % 4 == 0, "GemmArgs is copied word by word");
        const __attribute__((address_space(4))) uint32_t* s =
            reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(ka + offsetof(PPKSGroupArgs, p) + (size_t)q * sizeof(GemmArgs));
        uint32_t* d = reinterpret_cast<uint32_t*>(&P);
#pragma unroll
        for (int w = 0; w < (int)(sizeof(GemmArgs) / 4); ++w) d[w] = s[w];
        typedef __attribute__((address_space(1))) void gvoid_t;
        typedef __attribute__((address_space(1))) float gfloat_t;
        P.A = (const void*)(const gvoid_t*)P.A; P.B = (const void*)(const gvoid_t*)P.B; P.C = (void*)(gvoid_t*)P.C;
        P.bias = nullptr; P.resid = nullptr; P.aux = nullptr; P.partial = nullptr;
        P.colsum_a = (float*)(gfloat_t*)P.colsum_a;
    }
    ppks_body(P, wg, ga.tiles[q], split);
}

// ---- one reduce launch for the whole group: dW = (dW +) sum_z slab[z] in slice order, db = (db +) sum_z colsum[z]
struct DwReduceArgs {
    const float* slabs[POLUS_MAX_GROUP];   // [splits][n_out][n_in] or null (single slice wrote dW itself)
    const float* cs[POLUS_MAX_GROUP];      // [splits][n_out] or null
    float* dW[POLUS_MAX_GROUP];
    float* db[POLUS_MAX_GROUP];
    long lddw[POLUS_MAX_GROUP];
    int n_out[POLUS_MAX_GROUP], n_in[POLUS_MAX_GROUP], splits[POLUS_MAX_GROUP];
    int blk0[POLUS_MAX_GROUP + 1];         // first block of each problem
    int n, accumulate;
};

__global__ __launch_bounds__(256) void dw_group_reduce_kernel(DwReduceArgs ra) {
    const int b = blockIdx.x;
    int q = 0;
#pragma unroll
    for (int k = 1; k < POLUS_MAX_GROUP; ++k)
        if (k < ra.n && b >= ra.blk0[k]) q = k;
    // select the problem's fields with wave-uniform compares (no run-time indexing of kernel arguments)
    const float* slabs = nullptr; const float* cs = nullptr; float* dW = nullptr; float* db = nullptr;
    long lddw = 0; int n_out = 0, n_in = 0, splits = 0, blk0 = 0;
#pragma unroll
    for (int k = 0; k < POLUS_MAX_GROUP; ++k)
        if (k == q) { slabs = ra.slabs[k]; cs = ra.cs[k]; dW = ra.dW[k]; db = ra.db[k]; lddw = ra.lddw[k];
                      n_out = ra.n_out[k]; n_in = ra.n_in[k]; splits = ra.splits[k]; blk0 = ra.blk0[k]; }
    const long total = (long)n_out * n_in;
    const long quads = total / 4;                       // n_in % 4 == 0
    const long idx = (long)(b - blk0) * 256 + threadIdx.x;
    if (slabs != nullptr && idx < quads) {
        const long e = idx * 4;
        float4 s = *reinterpret_cast<const float4*>(slabs + e);
        for (int z = 1; z < splits; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(slabs + (long)z * total + e);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int m = (int)(e / n_in), n = (int)(e % n_in);
        float4* c = reinterpret_cast<float4*>(dW + (long)m * lddw + n);
        if (ra.accumulate) { const float4 o = *c; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *c = s;
    }
    // the bias gradient: the first blocks of the problem also take one column sum per thread
    if (cs != nullptr && db != nullptr && idx < n_out) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += cs[(long)z * n_out + idx];
        db[idx] = ra.accumulate ? db[idx] + s : s;
    }
}

}  // namespace

// Tiles a problem contributes to the grouped launch (256 x 256).
int polus_ppks_tiles(int n_out, int n_in) { return ((n_out + TM - 1) / TM) * ((n_in + TN - 1) / TN); }

// probs[k]: A = dY (K-strided), B = X (K-strided), C = slab base or dW, ldc, c_split_stride, k_per_split (multiple of
// 64), colsum_a or null, flags (ACCUM_C when a single slice writes dW); K % 64 == 0; M, N multiples of 8.
int polus_launch_gemm_ppks_grouped_dw(const GemmArgs* probs, int n, const int* splits, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_ppks_grouped_kernel;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_done = true;
    }
    PPKSGroupArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.n = n;
    int t0 = 0;
    for (int k = 0; k < n; ++k) {
        if (probs[k].K % TK != 0 || probs[k].k_per_split % TK != 0) return POLUS_ERR_INVALID;
        ga.p[k] = probs[k];
        ga.tiles[k] = polus_ppks_tiles(probs[k].M, probs[k].N);
        ga.tpad[k] = (ga.tiles[k] + 7) / 8 * 8;
        ga.wg0[k] = t0;
        t0 += ga.tpad[k] * splits[k];
    }
    ga.wg0[n] = t0;
    hipLaunchKernelGGL(kern, dim3(t0), dim3(NTHR), SMEM, st, ga);
    POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(ping-pong 256x256)");
    return POLUS_OK;
}

int polus_launch_dw_group_reduce(int n, const float* const* slabs, const float* const* cs, float* const* dW, float* const* db,
                                 const long* lddw, const int* n_out, const int* n_in, const int* splits, int accumulate,
                                 hipStream_t st) {
    DwReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.n = n; ra.accumulate = accumulate;
    int b0 = 0;
    for (int k = 0; k < n; ++k) {
        ra.slabs[k] = slabs[k]; ra.cs[k] = cs[k]; ra.dW[k] = dW[k]; ra.db[k] = db[k]; ra.lddw[k] = lddw[k];
        ra.n_out[k] = n_out[k]; ra.n_in[k] = n_in[k]; ra.splits[k] = splits[k];
        ra.blk0[k] = b0;
        long work = slabs[k] ? ((long)n_out[k] * n_in[k] / 4) : 0;
        if (cs[k] && db[k] && n_out[k] > work) work = n_out[k];
        b0 += (int)((work + 255) / 256);
    }
    ra.blk0[n] = b0;
    if (b0 == 0) return POLUS_OK;
    hipLaunchKernelGGL(dw_group_reduce_kernel, dim3(b0), dim3(256), 0, st, ra);
    POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(reduce)");
    return POLUS_OK;
}
