// Fused scaled-dot-product attention for BERT (head_dim 64, S <= 512 in practice) on MFMA.
//
// Forward: one workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16
// queries and sweeps the keys in blocks of 64 with an online softmax, so the S x S scores
// never reach HBM.  K and V blocks are staged [key][64] in LDS (row stride +32 B pad: the
// 160-B bf16 rows are conflict-free both for ds_read_b128 row reads and for
// ds_read_b64_tr_b16 column reads).
//   S^T = K Q^T is computed with the KEY on the MFMA row and the query on the column
//   (lane & 15): every lane then owns one query, its row max / row sum need two
//   cross-lane shuffles, and the probabilities are already the B operand of O^T = V^T P^T
//   (k = key) with no data movement: fragment element j of k-step s is key
//   16*(2s + (j>>2)) + 4g + (j&3), and the V^T fragment is read with the same key order by
//   the transposing LDS read.
// Backward recomputes P from the saved log-sum-exp in two kernels, both atomics-free and
// bitwise reproducible: dQ (same sweep as forward) and dK/dV (one workgroup per 64 keys
// sweeping the queries, dK^T/dV^T accumulators resident in registers).
// The additive mask is (1 - m) * -10000 exactly as polus/models.py:175-195 builds it.
#include "common.h"

namespace {

constexpr int D = 64;        // head dim
constexpr int BLK = 64;      // rows per LDS tile / per workgroup
constexpr float MASK_NEG = -10000.0f;
// exponentials run in base 2 (v_exp_f32 is 2^x): scale and key bias carry log2(e), one FMA per score
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16_t> { static constexpr int RS = 160, EPC = 8, LOG_CPR = 3, NLOAD = 2; };
template <> struct TileCfg<float> { static constexpr int RS = 288, EPC = 4, LOG_CPR = 4, NLOAD = 4; };

// 64 x 64 tile: rows row0.. of a [*, ld] matrix at column offset col0 -> LDS (zero fill past nrows)
// (NT threads load NT/4 rows: 64 rows for the 4-wave kernels, 16*NW for the wide ones)
template <typename T, int NT = 256>
__device__ __forceinline__ void tile_load(unsigned char* tile, const T* __restrict__ base, long ld, int row0,
                                          int nrows, int col0, int tid) {
    constexpr int LOG = TileCfg<T>::LOG_CPR;
#pragma unroll
    for (int k = 0; k < TileCfg<T>::NLOAD; ++k) {
        int c = tid + NT * k;
        int r = c >> LOG, ch = c & ((1 << LOG) - 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < nrows)
            v = *reinterpret_cast<const uint4*>(base + (long)(row0 + r) * ld + col0 + ch * TileCfg<T>::EPC);
        *reinterpret_cast<uint4*>(tile + r * TileCfg<T>::RS + ch * 16) = v;
    }
}

// row fragment straight from HBM (the per-wave resident operand): 8 consecutive d of row
template <typename T>
__device__ __forceinline__ void frag_global(Frag<T>& f, const T* __restrict__ p, bool valid);
template <>
__device__ __forceinline__ void frag_global<bf16_t>(Frag<bf16_t>& f, const bf16_t* __restrict__ p, bool valid) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid) v = *reinterpret_cast<const uint4*>(p);
    f.v = __builtin_bit_cast(bf16x8, v);
}
template <>
__device__ __forceinline__ void frag_global<float>(Frag<float>& f, const float* __restrict__ p, bool valid) {
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (valid) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

template <typename T>
__device__ __forceinline__ void frag_row(Frag<T>& f, const unsigned char* tile, int row, int sub, int g) {
    frag_load_row(f, tile + row * TileCfg<T>::RS + (sub * 32 + 8 * g) * (int)sizeof(T));
}

// transposed fragment: rows = features d0..d0+15 (lane i), k = tile rows kb + 16*(j>>2) + 4g + (j&3)
__device__ __forceinline__ void frag_tr(Frag<bf16_t>& f, const unsigned char* tile, int kb, int d0, int i, int g) {
    const unsigned char* p = tile + (kb + 4 * g + (i >> 2)) * 160 + (d0 + 4 * (i & 3)) * 2;
    s16x4 lo = lds_tr16(p);
    s16x4 hi = lds_tr16(p + 16 * 160);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ void frag_tr(Frag<float>& f, const unsigned char* tile, int kb, int d0, int i, int g) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        f.v[j] = *reinterpret_cast<const float*>(tile + (kb + 16 * (j >> 2) + 4 * g + (j & 3)) * 288 + (d0 + i) * 4);
}

// accumulator tiles (2s, 2s+1) -> B-operand fragment with the key order above
__device__ __forceinline__ void frag_from_acc(Frag<bf16_t>& f, const f32x4& t0, const f32x4& t1) {
    f.v[0] = (bf16_t)t0[0]; f.v[1] = (bf16_t)t0[1]; f.v[2] = (bf16_t)t0[2]; f.v[3] = (bf16_t)t0[3];
    f.v[4] = (bf16_t)t1[0]; f.v[5] = (bf16_t)t1[1]; f.v[6] = (bf16_t)t1[2]; f.v[7] = (bf16_t)t1[3];
}
__device__ __forceinline__ void frag_from_acc(Frag<float>& f, const f32x4& t0, const f32x4& t1) {
    f.v[0] = t0[0]; f.v[1] = t0[1]; f.v[2] = t0[2]; f.v[3] = t0[3];
    f.v[4] = t1[0]; f.v[5] = t1[1]; f.v[6] = t1[2]; f.v[7] = t1[3];
}

// reduce over the 4 lanes (g = 0..3) that share a column i
__device__ __forceinline__ float col_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float col_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

template <typename T>
__device__ __forceinline__ void store_acc_T(T* __restrict__ out, long ld, int row, int col0, const f32x4 (&o)[4],
                                            float mul, int g, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        float v[4] = {o[dt][0] * mul, o[dt][1] * mul, o[dt][2] * mul, o[dt][3] * mul};
        store4<T>(out + (long)row * ld + col0 + dt * 16 + 4 * g, v);
    }
}

struct AttnArgs {
    const void* qkv; const int32_t* mask; void* ctx; float* lse;
    const void* dctx; const float* delta; void* dqkv;
    int B, S, A, H;
    float scale;
    unsigned drop_thresh, drop_seed;   // attention-probability dropout; mask index ((b*A+h)*S+q)*S+key
    float drop_inv;
    const PolusDyn* dyn;               // per-step scalars in device memory (graph replay) or null
    int debug;                         // diagnostics (POLUS_ATTN_DEBUG): parts of the key-resident backward switched off
};

__device__ __forceinline__ float key_bias(const int32_t* mask, int b, int S, int key) {
    if (key >= S) return -INFINITY;
    if (!mask) return 0.0f;
    return (1.0f - (float)mask[(long)b * S + key]) * MASK_NEG;
}

// ---------------------------------------------------------------- forward
// NW waves per workgroup: 16*NW queries, and K / V are staged 16*NW keys at a time -- with NW = 16
// the whole key range of a 256-token sequence sits in LDS after ONE load + barrier pair (the
// 4-wave form pays a global-load latency and two barriers per 64 keys, which is what bounds it).
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(AttnArgs p) {
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    constexpr int QB = 16 * NW, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kc = smem;
    unsigned char* Vc = smem + QB * TileCfg<T>::RS;
    float* kbc = reinterpret_cast<float*>(smem + 2 * QB * TileCfg<T>::RS);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
    const int S = p.S, H = p.H;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const int q = q0 + wid * 16 + i;
    const bool qvalid = q < S;

    Frag<T> qf[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
        frag_global<T>(qf[sub], qkv + (long)q * ld + h * D + sub * 32 + 8 * g, qvalid);

    const float scale2 = p.scale * LOG2E;
    float m = -1e30f, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc0 = 0; kc0 < S; kc0 += QB) {
        __syncthreads();
        tile_load<T, NT>(Kc, qkv, ld, kc0, S, H + h * D, tid);
        tile_load<T, NT>(Vc, qkv, ld, kc0, S, 2 * H + h * D, tid);
        if (tid < QB) kbc[tid] = key_bias(p.mask, b, S, kc0 + tid) * LOG2E;
        __syncthreads();
      for (int kb0 = kc0; kb0 < kc0 + QB && kb0 < S; kb0 += BLK) {
        const unsigned char* Kt = Kc + (kb0 - kc0) * TileCfg<T>::RS;
        const unsigned char* Vt = Vc + (kb0 - kc0) * TileCfg<T>::RS;
        const float* kbias = kbc + (kb0 - kc0);

        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                Frag<T> a;
                frag_row<T>(a, Kt, kt * 16 + i, sub, g);
                mma16(s[kt], a, qf[sub]);  // D[key 4g+r][query i]
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][r] = fmaf(s[kt][r], scale2, kbias[kt * 16 + 4 * g + r]);   // log2 units
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = col_max(mx);
        const float m_new = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m_new); rs += s[kt][r]; }
        rs = col_sum(rs);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        if (p.drop_thresh) {   // the row sum above used the undropped probabilities (softmax first, then dropout)
            const unsigned rowb = (((unsigned)b * p.A + h) * S + (unsigned)q) * S + kb0 + 4 * g;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                float t4[4] = {s[kt][0], s[kt][1], s[kt][2], s[kt][3]};
                polus_dropout_run<4>(t4, p.drop_seed, rowb + kt * 16, p.drop_thresh, p.drop_inv, (S & 3) == 0);
                s[kt] = (f32x4){t4[0], t4[1], t4[2], t4[3]};
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> pb;
            frag_from_acc(pb, s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Frag<T> va;
                frag_tr(va, Vt, ks * 32, dt * 16, i, g);
                mma16(o[dt], va, pb);  // D[d 4g+r][query i]
            }
        }
      }
    }
    T* ctx = static_cast<T*>(p.ctx) + (long)b * S * H;
    store_acc_T<T>(ctx, H, q, h * D, o, 1.0f / l, g, qvalid);
    if (qvalid && g == 0) p.lse[((long)b * p.A + h) * S + q] = m * LN2 + __logf(l);
}

// ---------------------------------------------------------------- forward, bf16: LDS-DMA ring, 128 keys per softmax pass
// One workgroup = NW waves = 16 NW queries of one (batch, head).  K and V arrive by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, nothing returns to a VGPR, so the only waits in the kernel are the counted ones written below)
// into a ring of four 64-key blocks [K 64 x 128 B | V 64 x 128 B]; rows are packed (128 B) and the bank swizzle sits on
// the SOURCE address: LDS chunk pc of row r holds d-chunk pc ^ (r & 7) -- conflict-free for the ds_read_b128 row reads
// of K and for the ds_read_b64_tr_b16 column reads of V (the XOR is a lane constant: key block bases are multiples of 8).
// The DMA of the first four blocks (all of a 256-token sequence) is issued before anything is computed; block j's
// scores are computed behind a counted `vmcnt` that leaves the later blocks in flight, and a slot is refilled (longer
// sequences) as soon as both blocks of a pass have been consumed.  The online softmax advances 128 keys per pass: one
// row-max exchange per 128 keys instead of per 64, the row sums stay per lane until the end (the rescale factor is the
// same in the four lanes of a query).  Q and the key mask also come by LDS-DMA (Q into the fourth ring slot, which is
// refilled once the fragments are in registers); the output goes back through LDS so that HBM sees whole 128-byte rows.
// Same arithmetic per score as attn_fwd_kernel (scale, additive -10000 mask, base-2 exponentials, dropout after the
// row sum); the sums are taken in a different order, so results agree to rounding, not bit for bit.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
__device__ __forceinline__ void dma16(const void* g, unsigned char* lds) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lds_void_t*)lds, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* g, unsigned char* lds) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lds_void_t*)lds, 4, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// all but the `left * PPW` youngest LDS-DMA pieces of this wave have landed
template <int PPW> __device__ __forceinline__ void wait_blocks_left(int left) {
    if (left >= 3) wait_vm<3 * PPW>(); else if (left == 2) wait_vm<2 * PPW>(); else if (left == 1) wait_vm<PPW>(); else wait_vm<0>();
}
__device__ __forceinline__ float xor16(float v) {      // lane ^ 16 (inside a half wave): the swizzle unit, no LDS memory
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float col_max2(float v) {
    v = fmaxf(v, xor16(v));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float col_sum2(float v) {
    v += xor16(v);
    return v + __shfl_xor(v, 32, 64);
}

constexpr int FWD_RING = 4, FWD_STAGE = 2 * BLK * 128;   // ring slots; 16 KiB per 64-key block (K | V)
constexpr int FWD_MAX_S = 1024;                          // key-bias block in LDS: 4 B per key

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void attn_fwd_dma_kernel(AttnArgs p) {
    typedef bf16_t T;
    constexpr int QB = 16 * NW, NT = 64 * NW, PPW = 16 / NW;      // pieces (8 rows x 128 B) per wave and K/V block
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                                   // [4][K 8 KiB | V 8 KiB]
    float* kbc = reinterpret_cast<float*>(smem + FWD_RING * FWD_STAGE);   // [S rounded up to 64 NW] key bias, log2 units
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
    const int S = p.S, H = p.H;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const int q = q0 + wid * 16 + i;
    const int r8 = lane >> 3, src_chunk = ((lane & 7) ^ r8) * 8;  // this lane's row inside a piece, its d-chunk (elements)

    auto issue_block = [&](int j) {                               // 64 keys from j * 64 into ring slot j & 3
        unsigned char* st = ring + (j & 3) * FWD_STAGE;
#pragma unroll
        for (int e = 0; e < PPW; ++e) {
            const int piece = wid + e * NW;                       // 0-7: K rows 8 piece.., 8-15: V rows 8 (piece - 8)..
            const int row = min(j * BLK + (piece & 7) * 8 + r8, S - 1);
            dma16(qkv + (long)row * ld + (piece < 8 ? H : 2 * H) + h * D + src_chunk, st + piece * 1024);
        }
    };
    const int nblk = (S + BLK - 1) / BLK;
    // ---- prologue: the key mask of the whole sequence (ints for now), Q (into ring slot 3), blocks 0-2
    if (p.mask)
        for (int k0 = wid * 64; k0 < S; k0 += 64 * NW)
            dma4(p.mask + (long)b * S + min(k0 + lane, S - 1), reinterpret_cast<unsigned char*>(kbc) + k0 * 4);
    {
        unsigned char* qs = ring + 3 * FWD_STAGE;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int piece = wid + e * NW;
            dma16(qkv + (long)min(q0 + piece * 8 + r8, S - 1) * ld + h * D + src_chunk, qs + piece * 1024);
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) if (j < nblk) issue_block(j);
    wait_blocks_left<PPW>(min(nblk, 3));                          // mask and Q have landed
    __builtin_amdgcn_s_barrier();
    Frag<T> qf[2];
    {
        const int row = wid * 16 + i;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
            qf[sub].v = *reinterpret_cast<const bf16x8*>(ring + 3 * FWD_STAGE + row * 128 + (((sub * 4 + g) ^ (row & 7)) * 16));
    }
    for (int t = tid; t < nblk * BLK; t += NT) {                  // mask -> additive bias, in place (same thread reads and writes a word)
        float bias = 0.0f;
        if (p.mask) bias = (1.0f - (float)reinterpret_cast<const int*>(kbc)[t]) * MASK_NEG;
        kbc[t] = t < S ? bias * LOG2E : -INFINITY;
    }
    wait_lgkm();
    __builtin_amdgcn_s_barrier();                                 // Q fragments in registers everywhere: slot 3 is free
    if (nblk > 3) issue_block(3);
    int issued = min(nblk, 4);

    const float scale2 = p.scale * LOG2E;
    float m = -1e30f, l = 0.f;                                    // l: this lane's share of the row sum (4 lanes per query)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kx = i & 7;                                         // row & 7 of every K row this lane reads (rows 16 kt + i)
    const int k_off0 = i * 128 + ((g ^ kx) * 16), k_off1 = i * 128 + (((4 + g) ^ kx) * 16);
    // V^T fragment: rows 32 ks + 4 g + (i >> 2) (+16), d-chunk 2 dt + ((i & 3) >> 1), half (i & 1)
    const int vrow = 4 * g + (i >> 2), vx = vrow & 7;
    const int v_base = vrow * 128 + (i & 1) * 8, v_c = (i & 3) >> 1;

    for (int j0 = 0; j0 < nblk; j0 += 2) {
        const bool two = j0 + 1 < nblk;
        // ---- scores of the pass: S^T tiles [64 keys x 16 queries] per block, each behind its counted wait
        f32x4 s[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (jj == 0 || two) {
                wait_blocks_left<PPW>(issued - 1 - (j0 + jj));     // the later blocks stay in flight
                __builtin_amdgcn_s_barrier();
                const unsigned char* Kt = ring + ((j0 + jj) & 3) * FWD_STAGE;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    s[jj][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    Frag<T> a0, a1;
                    a0.v = *reinterpret_cast<const bf16x8*>(Kt + kt * 2048 + k_off0);
                    a1.v = *reinterpret_cast<const bf16x8*>(Kt + kt * 2048 + k_off1);
                    mma16(s[jj][kt], a0, qf[0]);                  // D[key 4g+r][query i]
                    mma16(s[jj][kt], a1, qf[1]);
                }
            } else {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[jj][kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        // ---- softmax step over the (up to) 32 scores of this lane; 4 lanes (g) share a query
        float mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            if (jj == 0 || two) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 kb = *reinterpret_cast<const f32x4*>(kbc + (j0 + jj) * BLK + kt * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[jj][kt][r] = fmaf(s[jj][kt][r], scale2, kb[r]);      // log2 units
                    mx = fmaxf(mx, fmaxf(fmaxf(s[jj][kt][0], s[jj][kt][1]), fmaxf(s[jj][kt][2], s[jj][kt][3])));
                }
            }
        mx = col_max2(mx);
        const float m_new = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        float rs = 0.f;
        Frag<T> pf[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[jj][kt][r] = __builtin_amdgcn_exp2f(s[jj][kt][r] - m_new); rs += s[jj][kt][r]; }
            if (p.drop_thresh) {   // the row sum uses the undropped probabilities (softmax first, then dropout)
                const unsigned rowb = (((unsigned)b * p.A + h) * S + (unsigned)q) * S + (j0 + jj) * BLK + 4 * g;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    float t4[4] = {s[jj][kt][0], s[jj][kt][1], s[jj][kt][2], s[jj][kt][3]};
                    polus_dropout_run<4>(t4, p.drop_seed, rowb + kt * 16, p.drop_thresh, p.drop_inv, (S & 3) == 0);
                    s[jj][kt] = (f32x4){t4[0], t4[1], t4[2], t4[3]};
                }
            }
            frag_from_acc(pf[jj][0], s[jj][0], s[jj][1]);
            frag_from_acc(pf[jj][1], s[jj][2], s[jj][3]);
        }
        l = l * alpha + rs;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        // ---- O^T += V^T P^T (a missing second block has P = 0: its slot holds finite leftovers or never-read bytes... skip it)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (jj == 0 || two) {
                const unsigned char* Vt = ring + ((j0 + jj) & 3) * FWD_STAGE + BLK * 128 + v_base;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const unsigned char* vp = Vt + ks * 4096 + (((2 * dt + v_c) ^ vx) * 16);
                        const s16x4 lo = lds_tr16(vp), hi = lds_tr16(vp + 16 * 128);
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        Frag<T> va;
                        va.v = __builtin_bit_cast(bf16x8, w);
                        mma16(o[dt], va, pf[jj][ks]);             // D[d 4g+r][query i]
                    }
            }
        }
        // ---- longer sequences: both slots of this pass are free once every wave is past its reads; refill them
        if (j0 + 4 < nblk) {
            wait_lgkm();
            __builtin_amdgcn_s_barrier();
            issue_block(j0 + 4);
            ++issued;
            if (j0 + 5 < nblk) { issue_block(j0 + 5); ++issued; }
        }
    }
    // ---- output: normalise, stage this wave's 16 x 64 tile in LDS (ring slot 0, wave-private 2 KiB, same chunk swizzle),
    // store whole 128-byte rows
    l = col_sum2(l);
    const float inv_l = 1.0f / l;
    wait_lgkm();
    __builtin_amdgcn_s_barrier();                                 // every wave is done reading K / V (and no DMA is in flight)
    unsigned char* ot = ring + wid * 2048;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        float v[4] = {o[dt][0] * inv_l, o[dt][1] * inv_l, o[dt][2] * inv_l, o[dt][3] * inv_l};
        store4<T>(reinterpret_cast<T*>(ot + i * 128 + (((2 * dt + (g >> 1)) ^ (i & 7)) * 16) + (g & 1) * 8), v);
    }
    wait_lgkm();
    T* ctx = static_cast<T*>(p.ctx) + (long)b * S * H;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 8 * t + r8, qq = q0 + wid * 16 + row;
        const uint4 v = *reinterpret_cast<const uint4*>(ot + row * 128 + (lane & 7) * 16);
        if (qq < S) *reinterpret_cast<uint4*>(ctx + (long)qq * H + h * D + (((lane & 7) ^ (row & 7)) * 8)) = v;
    }
    if (q < S && g == 0) p.lse[((long)b * p.A + h) * S + q] = m * LN2 + __logf(l);
}

// ---------------------------------------------------------------- dQ
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(AttnArgs p) {
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    constexpr int QB = 16 * NW, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kc = smem;
    unsigned char* Vc = smem + QB * TileCfg<T>::RS;
    float* kbc = reinterpret_cast<float*>(smem + 2 * QB * TileCfg<T>::RS);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
    const int S = p.S, H = p.H;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const T* dctx = static_cast<const T*>(p.dctx) + (long)b * S * H;
    const int q = q0 + wid * 16 + i;
    const bool qvalid = q < S;

    Frag<T> qf[2], dof[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        frag_global<T>(qf[sub], qkv + (long)q * ld + h * D + sub * 32 + 8 * g, qvalid);
        frag_global<T>(dof[sub], dctx + (long)q * H + h * D + sub * 32 + 8 * g, qvalid);
    }
    const long stat = ((long)b * p.A + h) * S + q;
    const float lse2 = (qvalid ? p.lse[stat] : 0.f) * LOG2E;
    const float scale2 = p.scale * LOG2E;
    // delta = rowsum(dO * O) of this query's head (softmax backward); computed here from the dO
    // fragments the lane already holds and published for the dK/dV kernel that runs next
    float dl = 0.f;
    {
        const T* ctx = static_cast<const T*>(p.ctx) + (long)b * S * H;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            Frag<T> of;
            frag_global<T>(of, ctx + (long)q * H + h * D + sub * 32 + 8 * g, qvalid);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)of.v[j] * (float)dof[sub].v[j];
        }
        dl = col_sum(dl);
        if (qvalid && g == 0) const_cast<float*>(p.delta)[stat] = dl;
    }

    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc0 = 0; kc0 < S; kc0 += QB) {
        __syncthreads();
        tile_load<T, NT>(Kc, qkv, ld, kc0, S, H + h * D, tid);
        tile_load<T, NT>(Vc, qkv, ld, kc0, S, 2 * H + h * D, tid);
        if (tid < QB) kbc[tid] = key_bias(p.mask, b, S, kc0 + tid) * LOG2E;
        __syncthreads();
      for (int kb0 = kc0; kb0 < kc0 + QB && kb0 < S; kb0 += BLK) {
        const unsigned char* Kt = Kc + (kb0 - kc0) * TileCfg<T>::RS;
        const unsigned char* Vt = Vc + (kb0 - kc0) * TileCfg<T>::RS;
        const float* kbias = kbc + (kb0 - kc0);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                Frag<T> a;
                frag_row<T>(a, Kt, kt * 16 + i, sub, g);
                mma16(s[kt], a, qf[sub]);
                frag_row<T>(a, Vt, kt * 16 + i, sub, g);
                mma16(dp[kt], a, dof[sub]);
            }
            if (p.drop_thresh) {
                float t4[4] = {dp[kt][0], dp[kt][1], dp[kt][2], dp[kt][3]};
                const unsigned idx0 = (((unsigned)b * p.A + h) * S + (unsigned)q) * S + kb0 + kt * 16 + 4 * g;
                polus_dropout_run<4>(t4, p.drop_seed, idx0, p.drop_thresh, p.drop_inv, (S & 3) == 0);
                dp[kt] = (f32x4){t4[0], t4[1], t4[2], t4[3]};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale2, kbias[kt * 16 + 4 * g + r]) - lse2);
                float dpe = dp[kt][r];
                s[kt][r] = pr * (dpe - dl) * p.scale;  // dS
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> dsb;
            frag_from_acc(dsb, s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Frag<T> ka;
                frag_tr(ka, Kt, ks * 32, dt * 16, i, g);
                mma16(dq[dt], ka, dsb);  // D[d][query]
            }
        }
      }
    }
    T* dqkv = static_cast<T*>(p.dqkv) + (long)b * S * ld;
    store_acc_T<T>(dqkv, ld, q, h * D, dq, 1.0f, g, qvalid);
}

// ---------------------------------------------------------------- dK, dV
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 1) void attn_bwd_dkv_kernel(AttnArgs p) {
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);   // bf16: 3 waves per SIMD (<= 168 registers)
    // Q and dO are staged QCH query rows per load + barrier pair (bf16: 128, i.e. two 64-row blocks)
    constexpr int QCH = sizeof(T) == 2 ? 2 * BLK : BLK;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * QCH * TileCfg<T>::RS + 2 * QCH * 4];
    unsigned char* Qc = smem;
    unsigned char* Oc = smem + QCH * TileCfg<T>::RS;
    float* lsec = reinterpret_cast<float*>(smem + 2 * QCH * TileCfg<T>::RS);
    float* deltac = lsec + QCH;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * BLK, h = blockIdx.y, b = blockIdx.z;
    const int S = p.S, H = p.H;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const T* dctx = static_cast<const T*>(p.dctx) + (long)b * S * H;
    const int key = k0 + wid * 16 + i;
    const bool kvalid = key < S;

    Frag<T> kf[2], vf[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        frag_global<T>(kf[sub], qkv + (long)key * ld + H + h * D + sub * 32 + 8 * g, kvalid);
        frag_global<T>(vf[sub], qkv + (long)key * ld + 2 * H + h * D + sub * 32 + 8 * g, kvalid);
    }
    const float kb2 = key_bias(p.mask, b, S, key) * LOG2E, scale2 = p.scale * LOG2E;
    const long stat0 = ((long)b * p.A + h) * S;

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    for (int qc0 = 0; qc0 < S; qc0 += QCH) {
        __syncthreads();
#pragma unroll
        for (int part = 0; part < QCH / BLK; ++part) {
            tile_load<T>(Qc + part * BLK * TileCfg<T>::RS, qkv, ld, qc0 + part * BLK, S, h * D, tid);
            tile_load<T>(Oc + part * BLK * TileCfg<T>::RS, dctx, H, qc0 + part * BLK, S, h * D, tid);
        }
        if (tid < QCH) {
            int qq = qc0 + tid;
            lsec[tid] = qq < S ? p.lse[stat0 + qq] * LOG2E : INFINITY;  // exp2(-inf) = 0 for padded queries
            deltac[tid] = qq < S ? p.delta[stat0 + qq] : 0.f;
        }
        __syncthreads();
      for (int qb0 = qc0; qb0 < qc0 + QCH && qb0 < S; qb0 += BLK) {
        const unsigned char* Qt = Qc + (qb0 - qc0) * TileCfg<T>::RS;
        const unsigned char* Ot = Oc + (qb0 - qc0) * TileCfg<T>::RS;
        const float* slse = lsec + (qb0 - qc0);
        const float* sdelta = deltac + (qb0 - qc0);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                Frag<T> a;
                frag_row<T>(a, Qt, qt * 16 + i, sub, g);
                mma16(s[qt], a, kf[sub]);   // D[query 4g+r][key i]
                frag_row<T>(a, Ot, qt * 16 + i, sub, g);
                mma16(dp[qt], a, vf[sub]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ql = qt * 16 + 4 * g + r;
                float pr = __builtin_amdgcn_exp2f(fmaf(s[qt][r], scale2, kb2) - slse[ql]);
                float dpe = dp[qt][r], pd = pr;
                if (p.drop_thresh) {
                    const unsigned idx = (((unsigned)b * p.A + h) * S + (unsigned)(qb0 + ql)) * S + (unsigned)key;
                    const bool keep = polus_keep(p.drop_seed, idx, p.drop_thresh);
                    dpe = keep ? dpe * p.drop_inv : 0.f;
                    pd = keep ? pr * p.drop_inv : 0.f;
                }
                dp[qt][r] = pr * (dpe - sdelta[ql]) * p.scale;  // dS
                s[qt][r] = pd;                                   // dropped P (what multiplied V forward)
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> pb, dsb;
            frag_from_acc(pb, s[2 * ks], s[2 * ks + 1]);
            frag_from_acc(dsb, dp[2 * ks], dp[2 * ks + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Frag<T> a;
                frag_tr(a, Ot, ks * 32, dt * 16, i, g);
                mma16(dv[dt], a, pb);   // dV^T[d][key] += dO^T P
                frag_tr(a, Qt, ks * 32, dt * 16, i, g);
                mma16(dk[dt], a, dsb);  // dK^T[d][key] += Q^T dS
            }
        }
      }
    }
    T* dqkv = static_cast<T*>(p.dqkv) + (long)b * S * ld;
    store_acc_T<T>(dqkv, ld, key, H + h * D, dk, 1.0f, g, kvalid);
    store_acc_T<T>(dqkv, ld, key, 2 * H + h * D, dv, 1.0f, g, kvalid);
}

// ---------------------------------------------------------------- backward in ONE pass (bf16, S in {64, 128, 256})
// One workgroup per (batch, head), S/16 waves, everything of the head on chip: Q and dO (all S rows) in LDS,
// K / V streamed in blocks of 32 keys (double-buffered).  Per key block:
//   phase 1  wave w owns queries 16w..16w+15 (as in the dQ kernel): S^T and dP^T tiles [32 keys x 16 queries] on
//            the matrix pipe, then -- ONCE per score -- the exponential, the dropout mask and dS; dQ^T += K^T dS^T
//            straight from the registers; the dropped probabilities P' and dS go to LDS as bf16 [query][key];
//   phase 2  the 16 output tiles of the block (dK^T and dV^T, [64 d x 32 keys] each) are dealt to the waves:
//            dK^T[d][key] = sum over ALL queries of Q^T[d][q] dS[q][key], dV^T = dO^T P', both operands by
//            transposing LDS reads with the same k (= query) order, final after S/32 MFMAs, stored at once.
// The two-kernel form evaluates every score twice (once per orientation of the S x S matrix: ~30 VALU
// lane-operations per score each time, against 7 matrix products it is VALU time that bounds it); here the
// second orientation costs an LDS round trip of two bf16 tiles instead.  Atomics-free, fixed summation order.
constexpr int KBLK = 32, RSS = 64;     // keys per block; bytes per query of the P' / dS staging tiles: two images (one per 16-key tile) of [S][32 B]

__device__ __forceinline__ void frag_tr_rs(Frag<bf16_t>& f, const unsigned char* tile, int rs, int kb, int c0, int i, int g) {
    const unsigned char* p = tile + (kb + 4 * g + (i >> 2)) * rs + (c0 + 4 * (i & 3)) * 2;
    s16x4 lo = lds_tr16(p);
    s16x4 hi = lds_tr16(p + 16 * rs);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, w);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_fused_kernel(AttnArgs p) {
    typedef bf16_t T;
    constexpr int S = 16 * NW, NT = 64 * NW, RS = TileCfg<T>::RS, NKB = S / KBLK, TPW = 16 / NW;
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qc = smem;
    unsigned char* Oc = Qc + S * RS;
    unsigned char* Kc = Oc + S * RS;                  // [2][KBLK][RS]
    unsigned char* Vc = Kc + 2 * KBLK * RS;           // [2][KBLK][RS]
    // P' / dS staging: [key tile 0-1][S queries][32 B = 16 keys], 8-byte granule g stored at g ^ ((q >> 2) & 3): the 8-byte
    // writes of a wave (16 queries x 4 granules) and the transposing reads (8 rows x 32 B) are both conflict-free (80-byte
    // rows cost a quarter of the LDS cycles of this kernel in bank conflicts: profiles/r02_attention_pmc.txt)
    unsigned char* Ps = Vc + 2 * KBLK * RS;           // [2][S][32]
    unsigned char* Ds = Ps + S * RSS;                 // [2][S][32]
    float* kbc = reinterpret_cast<float*>(Ds + S * RSS);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    const int H = p.H;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const T* dctx = static_cast<const T*>(p.dctx) + (long)b * S * H;
    T* dqkv = static_cast<T*>(p.dqkv) + (long)b * S * ld;
    const int q = wid * 16 + i;

    auto load_kv = [&](int kb, int buf) {          // 32 keys x (K | V) x 8 chunks of 16 B
        for (int c = tid; c < 2 * KBLK * 8; c += NT) {
            const int which = c >> 8, cc = c & 255, r = cc >> 3, ch = cc & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(qkv + (long)(kb * KBLK + r) * ld + (1 + which) * H + h * D + ch * 8);
            *reinterpret_cast<uint4*>((which ? Vc : Kc) + (buf * KBLK + r) * RS + ch * 16) = v;
        }
    };
    tile_load<T, NT>(Qc, qkv, ld, 0, S, h * D, tid);
    tile_load<T, NT>(Oc, dctx, H, 0, S, h * D, tid);
    load_kv(0, 0);
    if (tid < S) kbc[tid] = key_bias(p.mask, b, S, tid) * LOG2E;

    Frag<T> qf[2], dof[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        frag_global<T>(qf[sub], qkv + (long)q * ld + h * D + sub * 32 + 8 * g, true);
        frag_global<T>(dof[sub], dctx + (long)q * H + h * D + sub * 32 + 8 * g, true);
    }
    const long stat = ((long)b * p.A + h) * S + q;
    const float lse2 = p.lse[stat] * LOG2E;
    const float scale2 = p.scale * LOG2E;
    float dl = 0.f;                                   // delta = rowsum(dO * O) of this query's head
    {
        const T* ctx = static_cast<const T*>(p.ctx) + (long)b * S * H;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            Frag<T> of;
            frag_global<T>(of, ctx + (long)q * H + h * D + sub * 32 + 8 * g, true);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)of.v[j] * (float)dof[sub].v[j];
        }
        dl = col_sum(dl);
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int kb = 0; kb < NKB; ++kb) {
        const int cur = kb & 1;
        const unsigned char* Kt = Kc + cur * KBLK * RS;
        const unsigned char* Vt = Vc + cur * KBLK * RS;
        if (kb + 1 < NKB) load_kv(kb + 1, cur ^ 1);      // the other buffer was last read before the previous barrier
        // ---- phase 1
        f32x4 s[2], dp[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                Frag<T> a;
                frag_row<T>(a, Kt, kt * 16 + i, sub, g);
                mma16(s[kt], a, qf[sub]);                // D[key 4g+r][query i]
                frag_row<T>(a, Vt, kt * 16 + i, sub, g);
                mma16(dp[kt], a, dof[sub]);
            }
        }
        const unsigned rowb = (((unsigned)b * p.A + h) * S + (unsigned)q) * S + kb * KBLK + 4 * g;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float pd[4], dsv[4];
            bool keep[4] = {true, true, true, true};
            if (p.drop_thresh) {
                polus_keep4(p.drop_seed, rowb + kt * 16, p.drop_thresh, keep);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale2, kbc[kb * KBLK + kt * 16 + 4 * g + r]) - lse2);
                const float pk = keep[r] ? pr * p.drop_inv : 0.f;             // dropped P: what multiplied V forward
                const float dpe = keep[r] ? dp[kt][r] * p.drop_inv : 0.f;
                pd[r] = pk;
                dsv[r] = pr * (dpe - dl) * p.scale;
                s[kt][r] = dsv[r];
            }
            const int st_off = kt * (S * 32) + q * 32 + ((g ^ ((q >> 2) & 3)) * 8);
            store4<T>(reinterpret_cast<T*>(Ps + st_off), pd);
            store4<T>(reinterpret_cast<T*>(Ds + st_off), dsv);
        }
        {
            Frag<T> dsb;
            frag_from_acc(dsb, s[0], s[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Frag<T> ka;
                frag_tr(ka, Kt, 0, dt * 16, i, g);
                mma16(dq[dt], ka, dsb);                  // D[d][query]
            }
        }
        __syncthreads();
        // ---- phase 2: this block's dK^T / dV^T tiles, over all queries
#pragma unroll
        for (int e = 0; e < TPW; ++e) {
            const int t = wid * TPW + e;
            const int type = t >> 3, dt = (t >> 1) & 3, ktile = t & 1;
            const unsigned char* At = type ? Oc : Qc;
            const unsigned char* Bt = type ? Ps : Ds;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qs = 0; qs < S / 32; ++qs) {
                Frag<T> a, bb;
                frag_tr(a, At, qs * 32, dt * 16, i, g);                  // rows d, k = query
                {   // k = query (same order), columns = the 16 keys of image `ktile`; rows 32 qs + 4 g + (i >> 2) (+16): (row >> 2) & 3 = g
                    const unsigned char* bp = Bt + ktile * (S * 32) + (qs * 32 + 4 * g + (i >> 2)) * 32 + (((i & 3) ^ g) * 8);
                    const s16x4 lo = lds_tr16(bp), hi = lds_tr16(bp + 16 * 32);
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bb.v = __builtin_bit_cast(bf16x8, w);
                }
                mma16(acc, a, bb);                                        // D[d 4g+r][key i]
            }
            float v[4] = {acc[0], acc[1], acc[2], acc[3]};
            store4<T>(dqkv + (long)(kb * KBLK + ktile * 16 + i) * ld + (type ? 2 * H : H) + h * D + dt * 16 + 4 * g, v);
        }
        __syncthreads();
    }
    store_acc_T<T>(dqkv, ld, q, h * D, dq, 1.0f, g, true);
}

// ---- inline-asm LDS reads (invisible to the compiler's LDS-DMA alias waits) and a DPP quad broadcast
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef int v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_off(const unsigned char* p) { return (unsigned)(unsigned long)(const lds_u8*)p; }
// fragment = rows r..r+3 (elements 0-3) and r+16..r+19 (elements 4-7) of a transposing read; ROW16 = byte distance of 16 rows
template <int ROW16>
__device__ __forceinline__ void tr_pair_asm(Frag<bf16_t>& f, unsigned addr) {
    v2i_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(ROW16));
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const v4i_t w = {lo[0], lo[1], hi[0], hi[1]};
    f.v = __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ void ds_read128_asm(Frag<bf16_t>& f, unsigned addr) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    v4i_t w;
    asm volatile("ds_read_b128 %0, %1" : "=&v"(w) : "v"(addr));
    f.v = __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ uint2 ds_read64_asm(unsigned addr) {
    v2i_t w;
    asm volatile("ds_read_b64 %0, %1" : "=&v"(w) : "v"(addr));
    return make_uint2((unsigned)w[0], (unsigned)w[1]);
}
__device__ __forceinline__ void lds_fence() {            // results of the inline-asm reads above are in their registers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// value of lane (lane & ~3) + R of each quad of four lanes
template <int R> __device__ __forceinline__ unsigned quad_bcast(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, R | (R << 2) | (R << 4) | (R << 6), 0xF, 0xF, true);
}


// ---------------------------------------------------------------- backward in ONE pass, query-resident, 64-key blocks + LDS-DMA
// Same decomposition as attn_bwd_fused_kernel (one workgroup per (batch, head), S / 16 waves, phase 1 = scores, dS and dQ
// of the wave's 16 queries, phase 2 = the block's dK^T / dV^T tiles over all queries), rebuilt around what bounds that kernel
// -- two barriers and a register-staged global load per 32 keys, four LDS operand reads per phase-2 MFMA:
//   * key blocks of 64: half the barriers, phase 2 gives each wave tiles that SHARE their A operand (Q^T or dO^T), three
//     operand reads per MFMA instead of four;
//   * Q / dO images, K / V blocks by LDS-DMA into packed 128-byte rows (chunk-XOR swizzle on the source address, as in the
//     forward kernel); the K / V block is single-buffered: phase 2 does not read it, so block kb + 1 is issued right behind
//     the barrier that ends phase 1 and lands under phase 2 (every LDS read in here is inline asm -- a compiler-visible
//     read would be preceded by vmcnt(0), i.e. by the landing of the DMA just issued);
//   * P' / dS staging: four images (one per 16-key tile) of [S][32 B], conflict-free both ways.
// Arithmetic per score, dropout mask and summation order over the queries are those of attn_bwd_fused_kernel; dQ sums the
// keys in the same order as well: results are bit-identical to it.
constexpr int Q64_KB = 64;
template <int NW> struct Q64Cfg {
    static constexpr int S = 16 * NW;
    static constexpr int OFF_O = S * 128, OFF_K = 2 * S * 128, OFF_V = OFF_K + Q64_KB * 128, OFF_P = OFF_V + Q64_KB * 128,
                         OFF_D = OFF_P + 4 * S * 32, OFF_B = OFF_D + 4 * S * 32, SMEM = OFF_B + S * 4;
};

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void attn_bwd_q64_kernel(AttnArgs p) {
    typedef bf16_t T;
    typedef Q64Cfg<NW> C;
    constexpr int S = C::S, NT = 64 * NW, NKB = S / Q64_KB;
    constexpr int KVP = 16 / NW;                       // K / V pieces (8 rows x 128 B) per wave and block
    constexpr int COMBOS = NW >= 8 ? 1 : 8 / NW;       // (dK | dV, d-tile) pairs per wave in phase 2
    constexpr int KTW = NW == 16 ? 2 : 4;              // key tiles (of 16) per pair and wave
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qi = smem;
    unsigned char* Oi = smem + C::OFF_O;
    unsigned char* Kb = smem + C::OFF_K;
    unsigned char* Vb = smem + C::OFF_V;
    unsigned char* Ps = smem + C::OFF_P;               // [4][S][32]
    unsigned char* Ds = smem + C::OFF_D;
    float* kbc = reinterpret_cast<float*>(smem + C::OFF_B);
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x, b = blockIdx.y;
    const int H = p.H;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const T* dctx = static_cast<const T*>(p.dctx) + (long)b * S * H;
    T* dqkv = static_cast<T*>(p.dqkv) + (long)b * S * ld;
    const int q = wid * 16 + i;
    const int r8 = lane >> 3, src_chunk = ((lane & 7) ^ r8) * 8;

    auto issue_kv = [&](int kb) {
#pragma unroll
        for (int e = 0; e < KVP; ++e) {
            const int piece = wid + e * NW;            // 0-7: K rows, 8-15: V rows
            const long row = (long)(kb * Q64_KB + (piece & 7) * 8 + r8) * ld + h * D + src_chunk;
            dma16(qkv + row + (piece < 8 ? H : 2 * H), (piece < 8 ? Kb : Vb) + (piece & 7) * 1024);
        }
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {                      // Q and dO images: 2 NW pieces each, two per wave
        const int piece = wid + e * NW;
        dma16(qkv + (long)(piece * 8 + r8) * ld + h * D + src_chunk, Qi + piece * 1024);
        dma16(dctx + (long)(piece * 8 + r8) * H + h * D + src_chunk, Oi + piece * 1024);
    }
    issue_kv(0);
    if (tid < S) kbc[tid] = key_bias(p.mask, b, S, tid) * LOG2E;
    const long stat = ((long)b * p.A + h) * S + q;
    const float lse2 = p.lse[stat] * LOG2E;
    const float scale2 = p.scale * LOG2E;
    Frag<T> of[2];
    {
        const T* ctx = static_cast<const T*>(p.ctx) + (long)b * S * H;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) frag_global<T>(of[sub], ctx + (long)q * H + h * D + sub * 32 + 8 * g, true);
    }
    wait_vm<0>();
    wait_lgkm();
    __builtin_amdgcn_s_barrier();
    const int qoff0 = q * 128 + ((g ^ (q & 7)) * 16), qoff1 = q * 128 + (((4 + g) ^ (q & 7)) * 16);
    float dl = 0.f;                                    // delta = rowsum(dO * O) of this query's head
    {
        Frag<T> dof[2];
        ds_read128_asm(dof[0], lds_off(Oi + qoff0));
        ds_read128_asm(dof[1], lds_off(Oi + qoff1));
        lds_fence();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)of[sub].v[j] * (float)dof[sub].v[j];
        dl = col_sum(dl);
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane constants of the LDS reads (packed 128-byte rows, chunk c of row r at c ^ (r & 7))
    const int rr0 = i * 128 + ((g ^ (i & 7)) * 16), rr1 = i * 128 + (((4 + g) ^ (i & 7)) * 16);   // rows 16 kt + i, d-chunk 4 sub + g
    const int trow = 4 * g + (i >> 2), tx = trow & 7, thalf = (i & 1) * 8, tc = (i & 3) >> 1;    // transposing reads
    const int st_w = q * 32 + ((g ^ ((q >> 2) & 3)) * 8);                                        // staging write: image kt, row q
    const int st_r = trow * 32 + (((i & 3) ^ g) * 8);                                            // staging read: rows 32 qs + trow (+16)

    for (int kb = 0; kb < NKB; ++kb) {
        // ---- phase 1: my 16 queries x the 64 keys of the block (the query-side fragments are re-read from the resident images
        // every block: 16 registers that need not stay live across phase 2)
        f32x4 s[4];
        Frag<T> qf[2], dof[2];
        ds_read128_asm(qf[0], lds_off(Qi + qoff0));
        ds_read128_asm(qf[1], lds_off(Qi + qoff1));
        ds_read128_asm(dof[0], lds_off(Oi + qoff0));
        ds_read128_asm(dof[1], lds_off(Oi + qoff1));
        const unsigned rowb = (((unsigned)b * p.A + h) * S + (unsigned)q) * S + kb * Q64_KB + 4 * g;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {                        // one key tile at a time: its dP tile dies in the element-wise part
            Frag<T> ka[2], va[2];
            ds_read128_asm(ka[0], lds_off(Kb + kt * 2048 + rr0));
            ds_read128_asm(ka[1], lds_off(Kb + kt * 2048 + rr1));
            ds_read128_asm(va[0], lds_off(Vb + kt * 2048 + rr0));
            ds_read128_asm(va[1], lds_off(Vb + kt * 2048 + rr1));
            lds_fence();
            f32x4 dp = (f32x4){0.f, 0.f, 0.f, 0.f};
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma16(s[kt], ka[0], qf[0]);                         // D[key 4g+r][query i]
            mma16(s[kt], ka[1], qf[1]);
            mma16(dp, va[0], dof[0]);
            mma16(dp, va[1], dof[1]);
            float pd[4], dsv[4];
            bool keep[4] = {true, true, true, true};
            if (p.drop_thresh) polus_keep4(p.drop_seed, rowb + kt * 16, p.drop_thresh, keep);
            const f32x4 kbv = *reinterpret_cast<const f32x4*>(kbc + kb * Q64_KB + kt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale2, kbv[r]) - lse2);
                const float pk = keep[r] ? pr * p.drop_inv : 0.f;             // dropped P: what multiplied V forward
                const float dpe = keep[r] ? dp[r] * p.drop_inv : 0.f;
                pd[r] = pk;
                dsv[r] = pr * (dpe - dl) * p.scale;
                s[kt][r] = dsv[r];
            }
            store4<T>(reinterpret_cast<T*>(Ps + kt * (S * 32) + st_w), pd);
            store4<T>(reinterpret_cast<T*>(Ds + kt * (S * 32) + st_w), dsv);
        }
        {   // dQ^T += K^T dS^T over the 64 keys: K^T fragments by transposing reads, dS^T from the registers
            Frag<T> kT[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    tr_pair_asm<16 * 128>(kT[ks][dt], lds_off(Kb + (ks * 32 + trow) * 128 + (((2 * dt + tc) ^ tx) * 16) + thalf));
            lds_fence();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<T> dsb;
                frag_from_acc(dsb, s[2 * ks], s[2 * ks + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) mma16(dq[dt], kT[ks][dt], dsb);  // D[d][query]
            }
        }
        wait_lgkm();
        __builtin_amdgcn_s_barrier();
        // ---- the next block's K / V land under phase 2 (which reads neither)
        if (kb + 1 < NKB) issue_kv(kb + 1);
        // ---- phase 2: this block's dK^T / dV^T tiles over ALL queries; a wave's tiles share the A operand (Q^T or dO^T)
#pragma unroll
        for (int c = 0; c < COMBOS; ++c) {
            const int combo = NW == 16 ? (wid >> 1) : wid * COMBOS + c;       // type = combo >> 2 (0: dK, 1: dV), d-tile = combo & 3
            const int type = combo >> 2, dt = combo & 3, kt0 = NW == 16 ? (wid & 1) * 2 : 0;
            const unsigned char* At = type ? Oi : Qi;
            const unsigned char* Bt = type ? Ps : Ds;
            constexpr int QH = (S / 32) < 4 ? (S / 32) : 4;                     // query k-steps (of 32) per pass: 4 A + 4 B fragments in flight
            f32x4 acc[KTW];
#pragma unroll
            for (int k2 = 0; k2 < KTW; ++k2) acc[k2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q0 = 0; q0 < S / 32; q0 += QH) {
                Frag<T> af[QH];
#pragma unroll
                for (int qs = 0; qs < QH; ++qs)
                    tr_pair_asm<16 * 128>(af[qs], lds_off(At + ((q0 + qs) * 32 + trow) * 128 + (((2 * dt + tc) ^ tx) * 16) + thalf));
#pragma unroll
                for (int k2 = 0; k2 < KTW; ++k2) {
                    Frag<T> bf[QH];
#pragma unroll
                    for (int qs = 0; qs < QH; ++qs)
                        tr_pair_asm<16 * 32>(bf[qs], lds_off(Bt + (kt0 + k2) * (S * 32) + (q0 + qs) * 32 * 32 + st_r));
                    lds_fence();
#pragma unroll
                    for (int qs = 0; qs < QH; ++qs) mma16(acc[k2], af[qs], bf[qs]);      // D[d 4g+r][key i]
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < KTW; ++k2) {
                float v[4] = {acc[k2][0], acc[k2][1], acc[k2][2], acc[k2][3]};
                store4<T>(dqkv + (long)(kb * Q64_KB + (kt0 + k2) * 16 + i) * ld + (type ? 2 * H : H) + h * D + dt * 16 + 4 * g, v);
            }
        }
        // ---- block kb + 1 has landed (younger: this phase's COMBOS x KTW tile stores), every wave is done with the staging images
        if (kb + 1 < NKB) wait_vm<COMBOS * KTW>();
        wait_lgkm();
        __builtin_amdgcn_s_barrier();
    }
    store_acc_T<T>(dqkv, ld, q, h * D, dq, 1.0f, g, true);
}

// ---------------------------------------------------------------- backward in one pass, key-resident (bf16, S % 256 == 0)
// One workgroup = 8 waves = 256 keys of one (batch, head); wave w owns keys 32 w .. 32 w + 31 and keeps their dK^T and
// dV^T ([64 d x 32 keys] each) in registers while the workgroup sweeps the queries in slices of 32.  Per slice:
//   A  S = Q K^T and dP = dO V^T with the QUERY on the MFMA row and the key on the lane (K and V rows live in registers
//      as the column operands); the exponential, the dropout mask and dS once per score; the dropped probabilities P'
//      and dS are then ALREADY the column operands of dV^T += dO^T P' and dK^T += Q^T dS (contraction over the slice's
//      queries), so neither crosses LDS; only dS does, once, as bf16 [key][query];
//   B  dQ^T of the slice ([64 d x 32 q] = 8 tiles, one per wave) over all 256 keys: K^T fragments (registers, loaded
//      once) x dS^T (transposing LDS reads) -- final for this key block, no sum across waves.
// One barrier per slice: dS is double-buffered, dQ tiles are staged in LDS and leave as whole 128-byte rows one slice
// later.  Q and dO slices come by LDS-DMA two to three slices ahead behind counted waits (every transposing read is
// inline asm: the compiler would otherwise drain the prefetch with vmcnt(0) before each of them).
// Sequences longer than 256 keys take one workgroup per 256-key block: dK / dV are still final per workgroup; the dQ
// tiles go to an f32 slab per key block and attn_bwd_dq_finish_kernel adds the slabs in block order.
// The dropout mask is indexed ((b A + h) S + q) S + key with one hash per FOUR CONSECUTIVE KEYS; here a lane holds four
// queries of one key, so each lane hashes the quad of ONE of its queries and the four lanes of a key quad exchange the
// hashes by DPP -- the mask is bit-identical to the forward's.
constexpr int KR_KEYS = 256, KR_QS = 32, KR_RING = 4;
constexpr int KR_KV = KR_KEYS * 128;                 // K image / V image
constexpr int KR_SLICE = KR_QS * 128;                // a 32-query slice of Q or of dO
constexpr int KR_DS = 2 * KR_KEYS * 32;              // one dS buffer: [q-tile][key][16 q] bf16
constexpr int KR_OFF_RING = 2 * KR_KV, KR_OFF_DS = KR_OFF_RING + KR_RING * 2 * KR_SLICE, KR_OFF_DQ = KR_OFF_DS + 2 * KR_DS,
              KR_OFF_STAT = KR_OFF_DQ + 2 * KR_SLICE;   // + [S] lse (log2 units) + [S] delta

__global__ __launch_bounds__(512, 2) void attn_bwd_kres_kernel(AttnArgs p, float* __restrict__ dq_slabs) {
    typedef bf16_t T;
    constexpr int NT = 512;
    if (p.drop_thresh) p.drop_seed = polus_eff_seed(p.drop_seed, p.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kimg = smem;
    unsigned char* Vimg = smem + KR_KV;
    unsigned char* ring = smem + KR_OFF_RING;          // [4][Q slice 4 KiB | dO slice 4 KiB]
    unsigned char* dsb = smem + KR_OFF_DS;             // [2][2][256][32 B]
    unsigned char* dqb = smem + KR_OFF_DQ;             // [2][32 q][128 B]
    float* lsec = reinterpret_cast<float*>(smem + KR_OFF_STAT);
    const int S = p.S, H = p.H;
    float* deltac = lsec + S;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int key0 = kb * KR_KEYS;
    const long ld = 3L * H;
    const T* qkv = static_cast<const T*>(p.qkv) + (long)b * S * ld;
    const T* dctx = static_cast<const T*>(p.dctx) + (long)b * S * H;
    T* dqkv = static_cast<T*>(p.dqkv) + (long)b * S * ld;
    const int r8 = lane >> 3, src_chunk = ((lane & 7) ^ r8) * 8;
    const int nsl = S / KR_QS;

    // ---- LDS-DMA: K / V images of this key block (4 + 4 pieces per wave), then query slices 0-2 (one piece per wave and slice)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int piece = wid + 8 * e;                 // rows 8 piece ..
        const long row = (long)(key0 + piece * 8 + r8) * ld + h * D + src_chunk;
        dma16(qkv + row + H, Kimg + piece * 1024);
        dma16(qkv + row + 2 * H, Vimg + piece * 1024);
    }
    auto issue_slice = [&](int t) {                    // wave w: piece w of [Q rows 0-31 | dO rows 0-31] of slice t
        const int q = t * KR_QS + (wid & 3) * 8 + r8;
        unsigned char* dst = ring + (t & 3) * 2 * KR_SLICE + wid * 1024;
        if (wid < 4) dma16(qkv + (long)q * ld + h * D + src_chunk, dst);
        else dma16(dctx + (long)q * H + h * D + src_chunk, dst);
    };
#pragma unroll
    for (int t = 0; t < 3; ++t) if (t < nsl) issue_slice(t);
    // ---- per-query statistics: lse (log2 units) and delta = rowsum(dO * O); ordinary loads, consumed before the loop.
    // Whole rows per wave instruction (8 lanes x 16 B = one 128-byte head row, 8 rows per instruction), partial sums
    // folded over the 8 lanes of a row.
    {
        const T* ctx = static_cast<const T*>(p.ctx) + (long)b * S * H;
        const long stat0 = ((long)b * p.A + h) * S;
        for (int q8 = wid * 8; q8 < S; q8 += 256) {              // 64 rows per pass of the workgroup, 4 passes in flight
            bf16x8 o[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                          // every load first: one memory latency, not four
                const int q = q8 + 64 * u + r8;
                if (!(p.debug & 8)) {
                    o[u] = *reinterpret_cast<const bf16x8*>(ctx + (long)q * H + h * D + (lane & 7) * 8);
                    d[u] = *reinterpret_cast<const bf16x8*>(dctx + (long)q * H + h * D + (lane & 7) * 8);
                } else { o[u] = (bf16x8)(bf16_t)0.f; d[u] = o[u]; }
            }
            float l4[4];
            if ((lane & 7) == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) l4[u] = p.lse[stat0 + q8 + 64 * u + r8];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float dl = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)o[u][j] * (float)d[u][j];
                // the 8 lanes of a row: lane ^ 1, ^ 2, ^ 4 through the swizzle unit (no LDS memory)
                dl += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, dl), 0x041F));
                dl += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, dl), 0x081F));
                dl += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, dl), 0x101F));
                if ((lane & 7) == 0) { deltac[q8 + 64 * u + r8] = dl; lsec[q8 + 64 * u + r8] = l4[u] * LOG2E; }
            }
        }
    }
    float kb2[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) kb2[kt] = key_bias(p.mask, b, S, key0 + wid * 32 + kt * 16 + i) * LOG2E;
    wait_vm<0>();
    wait_lgkm();
    __builtin_amdgcn_s_barrier();

    // ---- operands that stay in registers: K / V rows of my 32 keys (column operands of S and dP), K^T of my dQ tile
    const int my_dt = wid & 3, my_qt = wid >> 2;
    Frag<T> kf[2][2], vf[2][2], ktr[8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int row = wid * 32 + kt * 16 + i, off = row * 128 + (((sub * 4 + g) ^ (row & 7)) * 16);
            ds_read128_asm(kf[kt][sub], lds_off(Kimg + off));
            ds_read128_asm(vf[kt][sub], lds_off(Vimg + off));
        }
    // transposing reads of a [rows][128 B] image with chunk swizzle: rows r0 + 4 g + (i >> 2) (+16), columns 16 dt + 4 (i & 3) ..
    const int trow = 4 * g + (i >> 2), tx = trow & 7, thalf = (i & 1) * 8, tc = (i & 3) >> 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        tr_pair_asm<16 * 128>(ktr[ks], lds_off(Kimg + (ks * 32 + trow) * 128 + (((2 * my_dt + tc) ^ tx) * 16) + thalf));
    lds_fence();
    // dS = P (dP - delta) / sqrt(d): 1/sqrt(d) = 0.125 is a power of two, so scaling the bf16 K^T operand of dQ (and the
    // dK^T sums at the end) instead of every dS is exact -- same bits, one multiply per score less
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) ktr[ks].v[j] = (bf16_t)((float)ktr[ks].v[j] * p.scale);

    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float scale2 = p.scale * LOG2E;
    // row reads of a slice: rows 16 qt + i, d-chunk 4 sub + g
    const int rr_off0 = i * 128 + ((g ^ (i & 7)) * 16), rr_off1 = i * 128 + (((4 + g) ^ (i & 7)) * 16);
    // dS staging: image qt, row = my key, granule g ^ ((row >> 2) & 3)
    int ds_w[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { const int row = wid * 32 + kt * 16 + i; ds_w[kt] = row * 32 + ((g ^ ((row >> 2) & 3)) * 8); }
    // dS^T reads (phase B): image my_qt, rows 32 ks + 4 g + (i >> 2) (+16), granule (i & 3) ^ ((row >> 2) & 3): (row >> 2) & 3 = g
    const int ds_r = my_qt * (KR_KEYS * 32) + trow * 32 + (((i & 3) ^ (g & 3)) * 8);
    const unsigned hash_row = ((unsigned)b * p.A + h) * S;      // + q, then * S + key
    const int my_r = i & 3;                                       // the query (of each group of four) whose quad this lane hashes

    f32x4 dq_prev = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nsl; ++t) {
        const unsigned char* sl = ring + (t & 3) * 2 * KR_SLICE;
        const int q0 = t * KR_QS;
        // ---- phase A
        Frag<T> qr[2][2], dor[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            ds_read128_asm(qr[qt][0], lds_off(sl + qt * 2048 + rr_off0));
            ds_read128_asm(qr[qt][1], lds_off(sl + qt * 2048 + rr_off1));
            ds_read128_asm(dor[qt][0], lds_off(sl + KR_SLICE + qt * 2048 + rr_off0));
            ds_read128_asm(dor[qt][1], lds_off(sl + KR_SLICE + qt * 2048 + rr_off1));
        }
        lds_fence();
        f32x4 s[2][2], dp[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                mma16(s[qt][kt], qr[qt][0], kf[kt][0]);          // D[query 4g+r][key i]
                mma16(s[qt][kt], qr[qt][1], kf[kt][1]);
                mma16(dp[qt][kt], dor[qt][0], vf[kt][0]);
                mma16(dp[qt][kt], dor[qt][1], vf[kt][1]);
            }
        // the transposed slice operands of dV^T / dK^T: issued now, consumed after the element-wise part
        Frag<T> qT[4], doT[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int off = trow * 128 + (((2 * dt + tc) ^ tx) * 16) + thalf;
            tr_pair_asm<16 * 128>(qT[dt], lds_off(sl + off));
            tr_pair_asm<16 * 128>(doT[dt], lds_off(sl + KR_SLICE + off));
        }
        Frag<T> pf[2], dsf[2];
        if (!(p.debug & 2))
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const f32x4 l2 = *reinterpret_cast<const f32x4*>(lsec + q0 + qt * 16 + 4 * g);
            const f32x4 de = *reinterpret_cast<const f32x4*>(deltac + q0 + qt * 16 + 4 * g);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                unsigned h1 = 0, h2 = 0;
                if (p.drop_thresh) {       // quad (query q0 + 16 qt + 4 g + my_r, keys 4 (key >> 2) ..): one hash per lane
                    const unsigned key = key0 + wid * 32 + kt * 16 + i;
                    const unsigned idx = (hash_row + (unsigned)(q0 + qt * 16 + 4 * g + my_r)) * (unsigned)S + (key & ~3u);
                    h1 = polus_hash32(p.drop_seed, idx >> 2);
                    h2 = polus_hash32_second(h1);
                }
                float pd[4], dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[qt][kt][r], scale2, kb2[kt]) - l2[r]);
                    float pk = pr, dpe = dp[qt][kt][r];
                    if (p.drop_thresh) {
                        // the hash of query 4 g + r sits in lane (lane & ~3) + r; my key is field (i & 3) of (h1, h2)
                        const unsigned a1 = r == 0 ? quad_bcast<0>(h1) : r == 1 ? quad_bcast<1>(h1) : r == 2 ? quad_bcast<2>(h1) : quad_bcast<3>(h1);
                        const unsigned a2 = r == 0 ? quad_bcast<0>(h2) : r == 1 ? quad_bcast<1>(h2) : r == 2 ? quad_bcast<2>(h2) : quad_bcast<3>(h2);
                        const unsigned hh = (i & 2) ? a2 : a1;
                        const bool keep = ((i & 1) ? (hh >> 16) : (hh & 0xFFFFu)) >= p.drop_thresh;
                        pk = keep ? pr * p.drop_inv : 0.f;
                        dpe = keep ? dpe * p.drop_inv : 0.f;
                    }
                    pd[r] = pk;
                    dsv[r] = pr * (dpe - de[r]);               // the 1/sqrt(d) factor: folded into K^T (dQ) and into the dK^T epilogue
                }
                s[qt][kt] = (f32x4){pd[0], pd[1], pd[2], pd[3]};
                dp[qt][kt] = (f32x4){dsv[0], dsv[1], dsv[2], dsv[3]};
                store4<T>(reinterpret_cast<T*>(dsb + (t & 1) * KR_DS + qt * (KR_KEYS * 32) + ds_w[kt]), dsv);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            frag_from_acc(pf[kt], s[0][kt], s[1][kt]);           // k = query: elements 0-3 <- q 4g.., 4-7 <- q 16 + 4g..
            frag_from_acc(dsf[kt], dp[0][kt], dp[1][kt]);
        }
        lds_fence();
        if (!(p.debug & 4))
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                mma16(dv[dt][kt], doT[dt], pf[kt]);              // dV^T[d][key] += dO^T[d][q] P'[q][key]
                mma16(dk[dt][kt], qT[dt], dsf[kt]);              // dK^T[d][key] += Q^T[d][q] dS[q][key]
            }
        // ---- slice t + 1 has landed (younger: the dQ store of slice t - 2 and the DMA of slice t + 2), then the one barrier
        {
            const int younger = (t >= 2 ? 1 : 0) + (t + 2 < nsl ? 1 : 0);
            if (younger >= 2) wait_vm<2>(); else if (younger == 1) wait_vm<1>(); else wait_vm<0>();
        }
        wait_lgkm();
        __builtin_amdgcn_s_barrier();
        // ---- dQ rows of slice t - 1 leave (staged in phase B of the previous iteration), the ring slot of slice t - 1 is refilled
        if (t >= 1 && dq_slabs) {
            // partial over this key block: f32 slab [kb][b S + q][H] (same program point as the bf16 rows below: the counted
            // waits above assume ONE store per wave and slice, issued here)
            float* dst = dq_slabs + ((long)kb * p.B * S + (long)b * S + q0 - KR_QS + my_qt * 16 + i) * H + h * D + my_dt * 16 + 4 * g;
            *reinterpret_cast<float4*>(dst) = make_float4(dq_prev[0], dq_prev[1], dq_prev[2], dq_prev[3]);
        }
        if (t >= 1 && !dq_slabs && !(p.debug & 16)) {
            const unsigned char* src = dqb + ((t - 1) & 1) * KR_SLICE;
            const int row = tid >> 4, c8 = tid & 15;                       // 32 rows x 16 chunks of 8 B
            const uint2 v = ds_read64_asm(lds_off(src + row * 128 + (((c8 >> 1) ^ (row & 7)) * 16) + (c8 & 1) * 8));
            lds_fence();
            *reinterpret_cast<uint2*>(dqkv + (long)(q0 - KR_QS + row) * ld + h * D + c8 * 4) = v;
        }
        if (t + 3 < nsl) issue_slice(t + 3);
        // ---- phase B: my dQ^T tile [16 d x 16 q] of slice t over the 256 keys of the block
        f32x4 dq = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(p.debug & 1)) {
            Frag<T> dst[8];
            const unsigned base = lds_off(dsb + (t & 1) * KR_DS + ds_r);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) tr_pair_asm<16 * 32>(dst[ks], base + ks * 32 * 32);
            lds_fence();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) mma16(dq, ktr[ks], dst[ks]);          // D[d 4g+r][query i]
        }
        dq_prev = dq;
        if (!dq_slabs) {
            float v[4] = {dq[0], dq[1], dq[2], dq[3]};
            const int row = my_qt * 16 + i;
            store4<T>(reinterpret_cast<T*>(dqb + (t & 1) * KR_SLICE + row * 128 + (((2 * my_dt + (g >> 1)) ^ (row & 7)) * 16) + (g & 1) * 8), v);
        }
    }
    // ---- epilogue: the last dQ slice, then dK^T / dV^T
    wait_lgkm();
    __builtin_amdgcn_s_barrier();
    if (dq_slabs) {
        float* dst = dq_slabs + ((long)kb * p.B * S + (long)b * S + (nsl - 1) * KR_QS + my_qt * 16 + i) * H + h * D + my_dt * 16 + 4 * g;
        *reinterpret_cast<float4*>(dst) = make_float4(dq_prev[0], dq_prev[1], dq_prev[2], dq_prev[3]);
    }
    if (!dq_slabs) {
        const unsigned char* src = dqb + ((nsl - 1) & 1) * KR_SLICE;
        const int row = tid >> 4, c8 = tid & 15;
        const uint2 v = *reinterpret_cast<const uint2*>(src + row * 128 + (((c8 >> 1) ^ (row & 7)) * 16) + (c8 & 1) * 8);
        *reinterpret_cast<uint2*>(dqkv + (long)((nsl - 1) * KR_QS + row) * ld + h * D + c8 * 4) = v;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const long key = key0 + wid * 32 + kt * 16 + i;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float a[4] = {dk[dt][kt][0] * p.scale, dk[dt][kt][1] * p.scale, dk[dt][kt][2] * p.scale, dk[dt][kt][3] * p.scale};
            float c[4] = {dv[dt][kt][0], dv[dt][kt][1], dv[dt][kt][2], dv[dt][kt][3]};
            store4<T>(dqkv + key * ld + H + h * D + dt * 16 + 4 * g, a);
            store4<T>(dqkv + key * ld + 2 * H + h * D + dt * 16 + 4 * g, c);
        }
    }
}

// dQ = sum over the key blocks of their f32 slabs, in block order (S > 256)
__global__ __launch_bounds__(256) void attn_bwd_dq_finish_kernel(const float* __restrict__ slabs, bf16_t* __restrict__ dqkv,
                                                                  long rows, int H, int nkb) {
    const long n4 = rows * (H / 4);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const long row = e / (H / 4);
        const int c = (int)(e - row * (H / 4)) * 4;
        float4 a = *reinterpret_cast<const float4*>(slabs + row * H + c);
        for (int k = 1; k < nkb; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + ((long)k * rows + row) * H + c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
        store4<bf16_t>(dqkv + row * 3L * H + c, v);
    }
}

int check_common(const char* who, int dtype, int B, int S, int A, int hd) {
    POLUS_REQUIRE(dtype == POLUS_F32 || dtype == POLUS_BF16, "%s: bad dtype %d", who, dtype);
    POLUS_REQUIRE(hd == D, "%s: head_dim must be 64 (got %d)", who, hd);
    POLUS_REQUIRE(B > 0 && S > 0 && A > 0, "%s: bad shape B=%d S=%d A=%d", who, B, S, A);
    POLUS_REQUIRE(B <= 65535 && A <= 65535, "%s: B and n_heads must be <= 65535", who);
    return POLUS_OK;
}

}  // namespace

namespace {
// Forward (WHICH = 0) and dQ (WHICH = 1) kernels: waves per workgroup by sequence length.  bf16:
// 8 waves (128 keys staged per load + barrier pair) from S >= 96, else 4; 16 waves (a whole
// 256-token sequence resident, 83 KiB) measures the same as 8 and is reachable with
// POLUS_ATTN_WAVES=16; f32 (288-byte rows, more registers): always 4.
template <int WHICH, typename T, int NW>
int launch_wide_nw(const AttnArgs& a, hipStream_t st) {
    auto kern = WHICH == 0 ? attn_fwd_kernel<T, NW> : attn_bwd_dq_kernel<T, NW>;
    constexpr int QB = 16 * NW;
    const size_t lds = 2 * (size_t)QB * TileCfg<T>::RS + QB * 4;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    dim3 grid((a.S + QB - 1) / QB, a.A, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a);
    return POLUS_OK;
}
template <int NW>
int launch_fwd_dma(const AttnArgs& a, hipStream_t st) {
    auto kern = attn_fwd_dma_kernel<NW>;
    const size_t lds = FWD_RING * FWD_STAGE + (size_t)((a.S + 64 * NW - 1) / (64 * NW)) * 64 * NW * 4;
    static bool attr_done = false;
    if (!attr_done) {       // once, for the longest sequence the kernel takes (the key-bias block grows with S)
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      FWD_RING * FWD_STAGE + (FWD_MAX_S + 64 * NW) * 4));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((a.S + 16 * NW - 1) / (16 * NW), a.A, a.B), dim3(64 * NW), lds, st, a);
    return POLUS_OK;
}
template <int NW>
int launch_q64(const AttnArgs& a, int n_heads, int B, hipStream_t st) {
    auto kern = attn_bwd_q64_kernel<NW>;
    static bool attr_done = false;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q64Cfg<NW>::SMEM));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_heads, B), dim3(64 * NW), Q64Cfg<NW>::SMEM, st, a);
    return POLUS_OK;
}
template <int NW>
int launch_fused(const AttnArgs& a, int n_heads, int B, size_t lds, hipStream_t st) {
    auto kern = attn_bwd_fused_kernel<NW>;
    static bool attr_done = false;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_heads, B), dim3(64 * NW), lds, st, a);
    return POLUS_OK;
}
template <int WHICH>
int launch_wide(int dtype, const AttnArgs& a, hipStream_t st) {
    if (dtype != POLUS_BF16) return launch_wide_nw<WHICH, float, 4>(a, st);
    if (a.S >= 96) return launch_wide_nw<WHICH, bf16_t, 8>(a, st);
    return launch_wide_nw<WHICH, bf16_t, 4>(a, st);
}
}  // namespace

extern "C" int polus_attention_fwd(int dtype, const void* qkv, const int32_t* mask, void* ctx, float* lse,
                                   int B, int S, int n_heads, int head_dim, float drop_p, uint32_t seed, void* stream) {
    int rc = check_common("polus_attention_fwd", dtype, B, S, n_heads, head_dim);
    if (rc) return rc;
    POLUS_REQUIRE(qkv && ctx && lse, "polus_attention_fwd: null pointer");
    POLUS_REQUIRE(polus_aligned16(qkv) && polus_aligned16(ctx), "polus_attention_fwd: pointers must be 16-byte aligned");
    AttnArgs a = {};
    POLUS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (long)B * n_heads * S * S < (1LL << 32), "polus_attention_fwd: bad dropout arguments");
    a.qkv = qkv; a.mask = mask; a.ctx = ctx; a.lse = lse;
    a.B = B; a.S = S; a.A = n_heads; a.H = n_heads * D; a.scale = 0.125f;
    a.drop_thresh = drop_p > 0.f ? polus_drop_thresh(drop_p) : 0u; a.drop_seed = seed; a.drop_inv = 1.0f / (1.0f - drop_p); a.dyn = polus_dyn();
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc2;
    if (dtype == POLUS_BF16 && S <= FWD_MAX_S)
        rc2 = S >= 96 ? launch_fwd_dma<8>(a, st) : launch_fwd_dma<4>(a, st);
    else
        rc2 = launch_wide<0>(dtype, a, st);
    if (rc2 != POLUS_OK) return rc2;
    POLUS_CHECK_LAUNCH("polus_attention_fwd");
    return POLUS_OK;
}

// delta [B, heads, S] for the two-kernel form; sequences of several 256-key blocks add one f32 dQ slab per block
// ([S / 256][B S][H], key-resident one-pass backward)
extern "C" size_t polus_attention_bwd_workspace_bytes(int B, int S, int n_heads) {
    size_t n = (size_t)B * S * n_heads * sizeof(float);
    if (S > KR_KEYS && S % KR_KEYS == 0 && S <= 2048) n += (size_t)(S / KR_KEYS) * B * S * n_heads * D * sizeof(float);      // (the key-resident kernel's range, polus_attention_bwd)
    return n;
}

extern "C" int polus_attention_bwd(int dtype, const void* qkv, const int32_t* mask, const void* ctx,
                                   const void* dctx, const float* lse, void* dqkv,
                                   int B, int S, int n_heads, int head_dim, float drop_p, uint32_t seed,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common("polus_attention_bwd", dtype, B, S, n_heads, head_dim);
    if (rc) return rc;
    POLUS_REQUIRE(qkv && ctx && dctx && lse && dqkv, "polus_attention_bwd: null pointer");
    POLUS_REQUIRE(polus_aligned16(qkv) && polus_aligned16(ctx) && polus_aligned16(dctx) && polus_aligned16(dqkv),
                  "polus_attention_bwd: pointers must be 16-byte aligned");
    size_t need = polus_attention_bwd_workspace_bytes(B, S, n_heads);
    if (!workspace || workspace_bytes < need) { polus_set_error("polus_attention_bwd: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    AttnArgs a = {};
    a.qkv = qkv; a.mask = mask; a.ctx = const_cast<void*>(ctx); a.lse = const_cast<float*>(lse); a.dctx = dctx; a.dqkv = dqkv;
    a.delta = static_cast<const float*>(workspace);
    a.B = B; a.S = S; a.A = n_heads; a.H = n_heads * D; a.scale = 0.125f;
    POLUS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "polus_attention_bwd: bad drop_p");
    a.drop_thresh = drop_p > 0.f ? polus_drop_thresh(drop_p) : 0u; a.drop_seed = seed; a.drop_inv = 1.0f / (1.0f - drop_p); a.dyn = polus_dyn();
    a.debug = polus_cfg().attn_debug;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((S + BLK - 1) / BLK, n_heads, B);
    // Which one-pass form: the key-resident kernel for sequences of several 256-key blocks (S = 512: 117 us against 134 us
    // for the two-kernel form at B = 16, 12 heads); at S = 256 the query-resident kernel below is faster (16 waves per CU
    // against 8: 82 us against 90-100 us at B = 64) unless POLUS_ATTN_BWD_KRES=2 forces the key-resident one (tests, A/B).
    const int kres = polus_cfg().attn_bwd_kres;
    if (dtype == POLUS_BF16 && polus_cfg().attn_fused && kres && S % KR_KEYS == 0 && S <= 2048 && (S > KR_KEYS || kres >= 2)) {
        // key-resident one-pass backward: one workgroup per (256-key block, head, batch)
        const int nkb = S / KR_KEYS;
        float* slabs = nkb > 1 ? static_cast<float*>(workspace) + (size_t)B * S * n_heads : nullptr;
        const size_t lds = KR_OFF_STAT + 2 * (size_t)S * sizeof(float);
        static bool attr_done = false;
        if (!attr_done) {
            POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kres_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_done = true;
        }
        hipLaunchKernelGGL(attn_bwd_kres_kernel, dim3(nkb, n_heads, B), dim3(512), lds, st, a, slabs);
        POLUS_CHECK_LAUNCH("polus_attention_bwd(key-resident)");
        if (nkb > 1) {
            hipLaunchKernelGGL(attn_bwd_dq_finish_kernel, dim3(1024), dim3(256), 0, st, slabs, static_cast<bf16_t*>(dqkv), (long)B * S, a.H, nkb);
            POLUS_CHECK_LAUNCH("polus_attention_bwd(dQ slabs)");
        }
        return POLUS_OK;
    }
    // query-resident one-pass forms: 64-key blocks by LDS-DMA at S = 256 (80.5 -> 76.8 us at 64 x 256, bit-identical), the 32-key-block
    // kernel at S = 64 / 128 (20.9 against 22.5 us at 32 x 128); the two were bit-identical where both ran (rounds 2-3)
    const int fused = polus_cfg().attn_fused;
    if (dtype == POLUS_BF16 && fused && S == 256) {
        // one pass, one workgroup per (batch, head), 64-key blocks by LDS-DMA
        int rc2 = launch_q64<16>(a, n_heads, B, st);
        if (rc2 != POLUS_OK) return rc2;
        POLUS_CHECK_LAUNCH("polus_attention_bwd(one pass, 64-key blocks)");
        return POLUS_OK;
    }
    if (dtype == POLUS_BF16 && fused && (S == 64 || S == 128)) {
        // one pass, one workgroup per (batch, head)
        const size_t lds = 2 * (size_t)S * TileCfg<bf16_t>::RS + 4 * (size_t)KBLK * TileCfg<bf16_t>::RS + 2 * (size_t)S * RSS + (size_t)S * 4;
        int rc2 = S == 128 ? launch_fused<8>(a, n_heads, B, lds, st) : launch_fused<4>(a, n_heads, B, lds, st);
        if (rc2 != POLUS_OK) return rc2;
        POLUS_CHECK_LAUNCH("polus_attention_bwd(fused)");
        return POLUS_OK;
    }
    if (dtype == POLUS_BF16) {
        { int rc2 = launch_wide<1>(dtype, a, st); if (rc2 != POLUS_OK) return rc2; }
        POLUS_CHECK_LAUNCH("polus_attention_bwd(dq)");
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<bf16_t>, grid, dim3(256), 0, st, a);
    } else {
        { int rc2 = launch_wide<1>(dtype, a, st); if (rc2 != POLUS_OK) return rc2; }
        POLUS_CHECK_LAUNCH("polus_attention_bwd(dq)");
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<float>, grid, dim3(256), 0, st, a);
    }
    POLUS_CHECK_LAUNCH("polus_attention_bwd(dkv)");
    return POLUS_OK;
}
