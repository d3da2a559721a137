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

// One 256 x 256 tile of dW over the contraction rows [kbeg, kend) (whole K-tiles).  `C` / `ldc`: where the tile's f32 values go,
// addressed by ABSOLUTE (m, n) -- a slab of the problem's shape, dW itself, or a dense tile slab offset by its origin;
// `csum_out`: this slice's row of column sums (bias gradient), indexed by absolute m, or null.
__device__ __forceinline__ void ppks_body(const GemmArgs& p, const int m0, const int n0, const int kbeg, const int kend,
                                          float* C, const long ldc, float* csum_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 2, wn = wid & 3;

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
    const bool want_colsum = csum_out != nullptr && n0 == 0;
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
            if (m < p.M) csum_out[m] = csum[e][0];
        }
    }
    GemmArgs q = p;
    q.C = C; q.ldc = ldc;
    epilogue_wave_128x64_lds<float, false>(q, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wid * 8704);
}

struct PPKSGroupArgs {
    GemmArgs p[POLUS_MAX_GROUP];
    int unit0[POLUS_MAX_GROUP + 1];   // first work unit of each problem, ascending; a problem's units = its K-slices x its tiles, slice-major
    int tiles[POLUS_MAX_GROUP];       // tiles of each problem
    int n;
};

// Work unit of workgroup b: the hardware deals workgroups to the 8 XCDs round-robin, and xcd_remap gives every XCD a contiguous
// run of the unit list -- consecutive tiles of one problem and one K-slice, which walk K in step and share their operand panels
// through that XCD's L2 -- of the SAME length on every XCD (+-1).  (Rounds 1-3 padded every problem's tile list to a multiple of
// 8 instead: 216 units then fell 32 / 30 / 30 / 28 / 24 / 24 / 24 / 24 on the XCDs.  A kernel on another stream is dealt to the
// XCDs round-robin as well and its dispatch stops at the first XCD without a free CU, so beside that layout the LayerNorm
// backward of the next layer did not start before the weight gradients had finished -- tools/exp/shadow_probe.py.)
__global__ __launch_bounds__(NTHR, 2) void gemm_ppks_grouped_kernel(PPKSGroupArgs ga) {
    const int unit = xcd_remap(blockIdx.x, gridDim.x);
    int q = 0;
#pragma unroll
    for (int k = 1; k < POLUS_MAX_GROUP; ++k)
        if (k < ga.n && unit >= ga.unit0[k]) q = k;
    const int rel = unit - ga.unit0[q];
    const int split = rel / ga.tiles[q], tile = rel - split * ga.tiles[q];
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
    const int tiles_n = (P.N + TN - 1) / TN;
    const int kbeg = split * P.k_per_split;
    ppks_body(P, (tile / tiles_n) * TM, (tile % tiles_n) * TN, kbeg, min(P.K, kbeg + P.k_per_split),
              static_cast<float*>(P.C) + (long)split * P.c_split_stride, P.ldc,
              P.colsum_a ? P.colsum_a + (long)split * P.M : nullptr);
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


// ---------------------------------------------------------------------------------------------------------------
// Stream-K hybrid of the grouped launch.  An encoder layer of BERT-base has 108 tiles; two K-slices each make 216 equal
// workgroups on 256 CUs -- 40 CUs idle for the whole launch.  Here every tile still gets `base` regular slices (same
// workgroup order as above, so the slices of one tile row walk K in step and share operands through their XCD's L2), but a
// regular slice is only `kr` K-tiles long, base x kr < T / 64, and the K-tiles a tile is left short of are dealt, in
// (tile, k) order, to the workgroups that are padding or absent above: each of those takes an equal run of that list
// -- parts of two to four tiles, one prologue / epilogue per part.  kr is chosen so that all 256 workgroups carry the
// same number of K-tiles (the remainder workgroups a few fewer: their extra epilogues).  Every part writes a dense
// 256 x 256 f32 slab, slot = position of the part in the tile's K order; dw_group_reduce_sk_kernel adds a tile's slots in
// that order, so the result does not depend on which CU ran what (bitwise reproducible, not bit-identical to the
// two-slice form: another summation tree).
struct PPKSSKArgs {
    GemmArgs p[POLUS_MAX_GROUP];
    int unit0[POLUS_MAX_GROUP + 1];   // first regular unit of each problem; a problem's regular units = base slices x its tiles, slice-major
    int tiles[POLUS_MAX_GROUP];
    int tile0[POLUS_MAX_GROUP + 1];   // first global tile of each problem (slab index)
    float* cs[POLUS_MAX_GROUP];       // column sums [slots][M] per problem, or null
    float* slabs;                     // [tiles of all problems][slots][256 x 256]
    int n, base, kr, krem;            // regular slices per tile, K-tiles per regular slice, K-tiles per tile left over
    int R, q_units, rem_units;        // remainder workgroups (units behind the regular ones); each takes q_units (+1 for the first rem_units) K-tiles of the left-over list
    int slots, n_regular;             // n_regular = unit0[n]
};

// remainder workgroup that owns unit u of the left-over list
__device__ __host__ __forceinline__ int sk_wg_of(int u, int q_units, int rem_units) {
    const int big = rem_units * (q_units + 1);
    return u < big ? u / (q_units + 1) : rem_units + (u - big) / q_units;
}

__global__ __launch_bounds__(NTHR, 2) void gemm_ppks_sk_kernel(PPKSSKArgs ga) {
    const int unit = xcd_remap(blockIdx.x, gridDim.x);    // the same number of workgroups on every XCD (see gemm_ppks_grouped_kernel)
    const int n_regular = ga.n_regular;
    const bool regular = unit < n_regular;
    int q = 0, tile = 0, kt0 = 0, kt1 = 0, slot = 0;      // current part: problem, tile in the problem, K-tiles [kt0, kt1), slab slot
    int u = 0, u1 = 0, r = 0;                             // remainder workgroup r: units [u, u1) of the left-over list
    if (regular) {
#pragma unroll
        for (int k = 1; k < POLUS_MAX_GROUP; ++k)
            if (k < ga.n && unit >= ga.unit0[k]) q = k;
        const int rel = unit - ga.unit0[q];
        const int split = rel / ga.tiles[q];
        tile = rel - split * ga.tiles[q];
        kt0 = split * ga.kr; kt1 = kt0 + ga.kr; slot = split;
    } else {
        r = unit - n_regular;
        if (r >= ga.R) return;
        u = r * ga.q_units + min(r, ga.rem_units);
        u1 = u + ga.q_units + (r < ga.rem_units ? 1 : 0);
    }
    typedef const __attribute__((address_space(4))) unsigned char* karg_t;
    karg_t ka = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        if (!regular) {
            const int j = u / ga.krem, off = u - j * ga.krem;
            const int len = min(ga.krem - off, u1 - u);
            q = 0;
#pragma unroll
            for (int k = 1; k < POLUS_MAX_GROUP; ++k)
                if (k < ga.n && j >= ga.tile0[k]) q = k;
            tile = j - ga.tile0[q];
            kt0 = ga.base * ga.kr + off; kt1 = kt0 + len;
            slot = ga.base + (r - sk_wg_of(j * ga.krem, ga.q_units, ga.rem_units));
            u += len;
        }
        GemmArgs P;
        float* cs = nullptr;
        {
            static_assert(sizeof(GemmArgs) % 4 == 0, "GemmArgs is copied word by word");
            const __attribute__((address_space(4))) uint32_t* sp =
                reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(ka + offsetof(PPKSSKArgs, p) + (size_t)q * sizeof(GemmArgs));
            uint32_t* d = reinterpret_cast<uint32_t*>(&P);
#pragma unroll
            for (int w = 0; w < (int)(sizeof(GemmArgs) / 4); ++w) d[w] = sp[w];
            typedef __attribute__((address_space(1))) void gvoid_t;
            P.A = (const void*)(const gvoid_t*)P.A; P.B = (const void*)(const gvoid_t*)P.B;
            P.C = nullptr; P.bias = nullptr; P.resid = nullptr; P.aux = nullptr; P.partial = nullptr; P.colsum_a = nullptr;
            P.flags = 0;
#pragma unroll
            for (int k = 0; k < POLUS_MAX_GROUP; ++k)
                if (k == q) cs = ga.cs[k];
        }
        const int tiles_n = (P.N + TN - 1) / TN;
        const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
        float* slab = ga.slabs + ((long)(ga.tile0[q] + tile) * ga.slots + slot) * (long)(TM * TN);
        ppks_body(P, m0, n0, kt0 * TK, kt1 * TK, slab - ((long)m0 * TN + n0), TN, cs ? cs + (long)slot * P.M : nullptr);
        if (regular || u >= u1) break;
        __syncthreads();                                   // the epilogue's LDS staging is read before the next part's operands land on it
    }
}

// dW = (dW +) sum of a tile's slab slots in K order; db likewise from the column sums of the tiles of the first tile column
struct DwReduceSKArgs {
    const float* slabs;
    const float* cs[POLUS_MAX_GROUP];
    float* dW[POLUS_MAX_GROUP];
    float* db[POLUS_MAX_GROUP];
    long lddw[POLUS_MAX_GROUP];
    int n_out[POLUS_MAX_GROUP], n_in[POLUS_MAX_GROUP];
    int tile0[POLUS_MAX_GROUP + 1];
    int n, accumulate, base, krem, q_units, rem_units, slots;
};

__global__ __launch_bounds__(256) void dw_group_reduce_sk_kernel(DwReduceSKArgs ra) {
    const int j = blockIdx.x >> 6, part = blockIdx.x & 63;        // 64 blocks of 4 rows x 256 columns per tile
    int q = 0;
#pragma unroll
    for (int k = 1; k < POLUS_MAX_GROUP; ++k)
        if (k < ra.n && j >= ra.tile0[k]) q = k;
    const float* cs = nullptr; float* dW = nullptr; float* db = nullptr;
    long lddw = 0; int n_out = 0, n_in = 0, t0 = 0;
#pragma unroll
    for (int k = 0; k < POLUS_MAX_GROUP; ++k)
        if (k == q) { cs = ra.cs[k]; dW = ra.dW[k]; db = ra.db[k]; lddw = ra.lddw[k]; n_out = ra.n_out[k]; n_in = ra.n_in[k]; t0 = ra.tile0[k]; }
    const int tiles_n = (n_in + TN - 1) / TN;
    const int tile = j - t0;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    // slots of tile j: the regular slices, then one per remainder workgroup that holds a part of its left-over K-tiles
    const int nslot = ra.base + sk_wg_of((j + 1) * ra.krem - 1, ra.q_units, ra.rem_units) - sk_wg_of(j * ra.krem, ra.q_units, ra.rem_units) + 1;
    const int row = part * 4 + (threadIdx.x >> 6), col = (threadIdx.x & 63) * 4;
    const float* src = ra.slabs + (long)j * ra.slots * (long)(TM * TN) + row * TN + col;
    if (m0 + row < n_out && n0 + col < n_in) {
        float4 s = *reinterpret_cast<const float4*>(src);
        for (int z = 1; z < nslot; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(src + (long)z * (TM * TN));
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float4* c = reinterpret_cast<float4*>(dW + (long)(m0 + row) * lddw + n0 + col);
        if (ra.accumulate) { const float4 o = *c; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *c = s;
    }
    if (cs != nullptr && db != nullptr && n0 == 0 && part == 0) {
        const int m = m0 + threadIdx.x;
        if (m < n_out) {
            float s = 0.f;
            for (int z = 0; z < nslot; ++z) s += cs[(long)z * n_out + m];
            db[m] = ra.accumulate ? db[m] + s : s;
        }
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
    int u0 = 0;
    for (int k = 0; k < n; ++k) {
        if (probs[k].K % TK != 0 || probs[k].k_per_split % TK != 0) return POLUS_ERR_INVALID;
        ga.p[k] = probs[k];
        ga.tiles[k] = polus_ppks_tiles(probs[k].M, probs[k].N);
        ga.unit0[k] = u0;
        u0 += ga.tiles[k] * splits[k];
    }
    ga.unit0[n] = u0;
    hipLaunchKernelGGL(kern, dim3(u0), dim3(NTHR), SMEM, st, ga);
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

// ---- stream-K hybrid (see gemm_ppks_sk_kernel).  The plan is a pure function of the shapes and the CU count, so the
// workspace query and the launch agree.
int polus_ppks_sk_plan(const int* n_out, const int* n_in, int n, int T, int ncu, int delta, PPKSSKPlan* pl) {
    memset(pl, 0, sizeof(*pl));
    if (n < 1 || n > POLUS_MAX_GROUP || T % TK != 0 || ncu < 8) return 0;
    const int nkt = T / TK;
    int ttot = 0;
    for (int k = 0; k < n; ++k) {
        pl->tiles[k] = polus_ppks_tiles(n_out[k], n_in[k]);
        pl->tile0[k] = ttot;
        ttot += pl->tiles[k];
    }
    pl->tile0[n] = ttot;
    const int base = ncu / ttot;
    if (base < 1) return 0;
    const int R = ncu - base * ttot;
    if (R < 4) return 0;                              // (next to) no CU idle: the even split is as good
    long units = (long)ttot * nkt;
    int kr = (int)((units + ncu - 1) / ncu) + delta;
    if (kr < 4 || base * kr >= nkt) return 0;
    const int krem = nkt - base * kr;
    const long rem_total = (long)ttot * krem;
    if (rem_total < R) return 0;
    pl->n = n; pl->base = base; pl->kr = kr; pl->krem = krem; pl->R = R;
    pl->q_units = (int)(rem_total / R); pl->rem_units = (int)(rem_total % R);
    // parts a tile's left-over K-tiles can fall into: ceil(krem / q) + 1 workgroups at most
    pl->slots = base + (krem + pl->q_units - 1) / pl->q_units + 1;
    int w = 0;
    for (int k = 0; k < n; ++k) { pl->unit0[k] = w; w += pl->tiles[k] * base; }
    pl->unit0[n] = w;
    pl->grid = w + R;
    pl->ttot = ttot;
    return 1;
}

int polus_launch_gemm_ppks_sk(const GemmArgs* probs, const PPKSSKPlan& pl, float* slabs, float* const* cs, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm_ppks_sk_kernel;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_done = true;
    }
    PPKSSKArgs ga;
    memset(&ga, 0, sizeof(ga));
    for (int k = 0; k < pl.n; ++k) {
        if (probs[k].K % TK != 0) return POLUS_ERR_INVALID;
        ga.p[k] = probs[k];
        ga.unit0[k] = pl.unit0[k]; ga.tiles[k] = pl.tiles[k]; ga.tile0[k] = pl.tile0[k];
        ga.cs[k] = cs[k];
    }
    ga.unit0[pl.n] = pl.unit0[pl.n]; ga.tile0[pl.n] = pl.tile0[pl.n];
    ga.slabs = slabs;
    ga.n = pl.n; ga.base = pl.base; ga.kr = pl.kr; ga.krem = pl.krem; ga.R = pl.R; ga.q_units = pl.q_units; ga.rem_units = pl.rem_units;
    ga.slots = pl.slots; ga.n_regular = pl.unit0[pl.n];
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(NTHR), SMEM, st, ga);
    POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(ping-pong 256x256, stream-K remainder)");
    return POLUS_OK;
}

int polus_launch_dw_group_reduce_sk(const PPKSSKPlan& pl, const float* slabs, const float* const* cs, float* const* dW, float* const* db,
                                    const long* lddw, const int* n_out, const int* n_in, int accumulate, hipStream_t st) {
    DwReduceSKArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.slabs = slabs;
    for (int k = 0; k < pl.n; ++k) {
        ra.cs[k] = cs[k]; ra.dW[k] = dW[k]; ra.db[k] = db[k]; ra.lddw[k] = lddw[k]; ra.n_out[k] = n_out[k]; ra.n_in[k] = n_in[k];
        ra.tile0[k] = pl.tile0[k];
    }
    ra.tile0[pl.n] = pl.tile0[pl.n];
    ra.n = pl.n; ra.accumulate = accumulate; ra.base = pl.base; ra.krem = pl.krem; ra.q_units = pl.q_units; ra.rem_units = pl.rem_units;
    ra.slots = pl.slots;
    hipLaunchKernelGGL(dw_group_reduce_sk_kernel, dim3(pl.ttot * 64), dim3(256), 0, st, ra);
    POLUS_CHECK_LAUNCH("polus_dense_bwd_params_grouped(reduce, stream-K slabs)");
    return POLUS_OK;
}
