// Shared device/host helpers for the Polus MI355X (gfx950) kernel library.
// Wave = 64 lanes everywhere; MFMA shape 16x16 (bf16: K=32, f32: K=4 x 8 issues).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/polus_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- error plumbing (host)
void polus_set_error(const char* fmt, ...);
#define POLUS_FAIL(...) do { polus_set_error(__VA_ARGS__); return POLUS_ERR_INVALID; } while (0)
#define POLUS_REQUIRE(cond, ...) do { if (!(cond)) { polus_set_error(__VA_ARGS__); return POLUS_ERR_INVALID; } } while (0)
#define POLUS_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) { polus_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); return POLUS_ERR_HIP; } } while (0)
#define POLUS_HIP(call) do { hipError_t e__ = (call); \
    if (e__ != hipSuccess) { polus_set_error("%s failed: %s", #call, hipGetErrorString(e__)); return POLUS_ERR_HIP; } } while (0)

static inline size_t polus_dtype_size(int dt) { return dt == POLUS_BF16 ? 2 : 4; }
static inline bool polus_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- tuning knobs (host)
// The POLUS_* environment switches (A/B runs, tests) are read ONCE at library load -- a getenv per
// launch sat on the launch path of every GEMM -- and again only through polus_reload_env().
struct PolusCfg {
    int gemm_pp;           // POLUS_GEMM_PP: -1 off, 0 per-shape choice (default), 256 / 192 force that tile where legal
    int gemm_v1;           // POLUS_GEMM_V1: 128x128 register-staged kernel for everything
    int dw_ungrouped;      // POLUS_DW_UNGROUPED: one dW launch per matrix
    int dw_fused_reduce;   // POLUS_DW_FUSED_REDUCE: 1 (default) one reduce launch per grouped dW
    int ln_bwd_blocks;     // POLUS_LN_BWD_BLOCKS: cap on LayerNorm-backward workgroups (default 512 = all resident at once; 1024 = round-1 grid)
    int ln_fin_single;     // POLUS_LN_FIN_SINGLE: LayerNorm-backward partials up to this many rows are reduced by one finalize launch, more in two stages (default 512)
    int gemm_ring128;      // POLUS_GEMM_RING128: -1 never, 0 (default) where the heuristic picks it, 1 wherever it applies (bf16 C, K-contiguous operands)
    int gemm_auto_split;   // POLUS_GEMM_AUTO_SPLIT: 1 (default) polus_gemm_auto_split recommends K slices for under-filled bf16 Dense GEMMs; 0 = always 1
    int ln_halfwave;       // POLUS_LN_HALFWAVE: 1 (default) half-wave-per-row LayerNorm kernels with 16-byte accesses (bf16, H % 256 == 0)
    int gemm_order;        // POLUS_GEMM_ORDER: column tiles an XCD's concurrent ping-pong tiles span (0 = row-major run; default 4)
    int gemm_persist;      // POLUS_GEMM_PERSIST: 1 (default) = the multi-round 256-wide launches as one persistent workgroup per CU (next tile's prologue under the epilogue), 2 = every multi-round ping-pong launch, 0 = never
    int reserve_cus;       // POLUS_GEMM_RESERVE_CUS: CUs the tile-shape choice leaves to concurrent RCCL channel kernels (default 0)
    int attn_bwd_kres;     // POLUS_ATTN_BWD_KRES: 1 (default) key-resident one-pass attention backward for bf16 sequences of several 256-key blocks, 2 = also at S = 256, 0 = never
    int attn_debug;        // POLUS_ATTN_DEBUG: diagnostics, parts of the key-resident attention backward switched off (wrong results)
    int dw_streamk;        // POLUS_DW_STREAMK: grouped dW with a stream-K remainder on the CUs the even K split leaves idle
    int dw_sk_cus;         // POLUS_DW_SK_CUS: CUs the stream-K grouped dW launch is planned for (0 = all that are not reserved)
    int dw_sk_delta;       // POLUS_DW_SK_DELTA: K-tiles a regular slice carries more than the even share (the remainder workgroups' extra epilogues)
    int attn_fused;        // POLUS_ATTN_FUSED: 1 (default) one-pass attention backward (bf16): 64-key blocks by LDS-DMA at S = 256, 32-key blocks at S = 64 / 128, key-resident from S = 512; 0 = two kernels
};
const PolusCfg& polus_cfg();
int polus_num_cus();        // CUs of the current device
int polus_reserved_cus();   // cfg.reserve_cus while the reserve is switched on (polus_set_reserve_active), else 0

// ---------------------------------------------------------------- per-step scalars in device memory
// A captured HIP graph replays the SAME kernel arguments every step, so what changes from step to step --
// the dropout seeds, the learning rate -- must come from memory.  polus_set_dynamic_params() registers a
// device block {salt, lr, lr_t, 0}; while it is set every launcher passes it on, and a kernel computes
//     seed_eff = mix(seed + salt),  mix(x) = (x ^ x >> 15) * 0x2C1B3C6D
// -- the host computes exactly mix(linear seed + step term) when no block is registered
// (polus_amd/models.py dropout_seed), so eager and graph-replayed steps draw the same masks.
struct PolusDyn { uint32_t salt; float lr, lr_t; uint32_t pad; };
const PolusDyn* polus_dyn();          // the registered block (device pointer) or null
__device__ __forceinline__ uint32_t polus_eff_seed(uint32_t seed, const PolusDyn* dyn) {
    if (!dyn) return seed;
    uint32_t x = seed + dyn->salt;
    x ^= x >> 15;
    return x * 0x2C1B3C6Du;
}

// ---------------------------------------------------------------- scalar conversions
template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

// 4-element vector load/store of T as floats (bf16: 8 B, f32: 16 B)
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    bf16x4 t; t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = t;
}

// ---------------------------------------------------------------- activations
// erf: Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, one v_rcp + one v_exp + 6 FMAs) instead of
// the ~60-instruction libm erff: the GELU epilogue of FFN1 touches 50 M elements per layer.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float r = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float swish_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float swish_grad_f(float x) {
    float s = 1.0f / (1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
// bf16 engine: GELU and its derivative without transcendentals.  Phi(x) = 0.5 + x q(u) and
// gelu'(x) = Phi(x) + x phi(x) = 0.5 + x w(u) with u = 2 x^2 / c^2 - 1 on |x| <= c = 4.5 (clamped
// outside: Phi(4.5) = 1 - 3.4e-6), q and w degree-10 minimax fits evaluated by Horner on pairs
// (v_pk_fma_f32).  |gelu error| <= 2e-5, |gelu' error| <= 1e-4 -- far below a bf16 ulp of the
// outputs they feed; the erf/exp forms above cost ~3x as many VALU cycles, and epilogue VALU time is
// not hidden behind the matrix pipe.  The f32 engine keeps erf_fast.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t x, const float (&c)[11]) {
    f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f)};
    f32x2_t s = xc * xc;
    f32x2_t u = __builtin_elementwise_fma(s, (f32x2_t){2.0f / 20.25f, 2.0f / 20.25f}, (f32x2_t){-1.0f, -1.0f});
    f32x2_t r = {c[10], c[10]};
#pragma unroll
    for (int k = 9; k >= 0; --k) r = __builtin_elementwise_fma(r, u, (f32x2_t){c[k], c[k]});
    return __builtin_elementwise_fma(xc, r, (f32x2_t){0.5f, 0.5f});
}
__device__ __forceinline__ f32x2_t gelu_cdf2(f32x2_t x) {
    const float q[11] = {1.569049305e-01f, -7.719386095e-02f, 5.470119065e-02f, -4.010921159e-02f, 2.828393083e-02f,
                         -1.902094059e-02f, 1.143922652e-02f, -5.251548121e-03f, 2.716277400e-03f, -2.340015935e-03f,
                         9.806926012e-04f};
    return gelu_poly2(x, q);
}
__device__ __forceinline__ f32x2_t gelu_grad2(f32x2_t x) {
    const float w[11] = {1.594289755e-01f, -9.003904569e-02f, 8.714168751e-02f, -9.350841452e-02f, 9.645831298e-02f,
                         -9.557466143e-02f, 7.437658385e-02f, -3.356380937e-02f, 2.252388123e-02f, -3.068672777e-02f,
                         1.457471506e-02f};
    return gelu_poly2(x, w);
}
// `act` is wave-uniform (a kernel argument): real branches, so only one activation is evaluated
// (nested selects would compute erf, exp and tanh for every element).
template <int N, bool FAST = false> __device__ __forceinline__ void apply_act_n(int act, float (&v)[N]) {
    if (act == POLUS_ACT_GELU) {
        if (FAST && (N % 2) == 0) {
#pragma unroll
            for (int r = 0; r < N; r += 2) {
                f32x2_t x = {v[r], v[r + 1]};
                f32x2_t y = x * gelu_cdf2(x);
                v[r] = y[0]; v[r + 1] = y[1];
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = gelu_f(v[r]);
    } else if (act == POLUS_ACT_SWISH) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = swish_f(v[r]);
    } else if (act == POLUS_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = fmaxf(v[r], 0.0f);
    } else if (act == POLUS_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = tanhf(v[r]);
    }
}
// v[r] *= act'(u[r]) given the pre-activations u
template <int N, bool FAST = false> __device__ __forceinline__ void apply_act_grad_n(int act, float (&v)[N], const float (&u)[N]) {
    if (act == POLUS_ACT_GELU) {
        if (FAST && (N % 2) == 0) {
#pragma unroll
            for (int r = 0; r < N; r += 2) {
                f32x2_t d = (f32x2_t){v[r], v[r + 1]} * gelu_grad2((f32x2_t){u[r], u[r + 1]});
                v[r] = d[0]; v[r + 1] = d[1];
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] *= gelu_grad_f(u[r]);
    } else if (act == POLUS_ACT_SWISH) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] *= swish_grad_f(u[r]);
    } else if (act == POLUS_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = u[r] > 0.0f ? v[r] : 0.0f;
    } else if (act == POLUS_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < N; ++r) { float t = tanhf(u[r]); v[r] *= 1.0f - t * t; }
    }
}

// ---------------------------------------------------------------- dropout (counter-based)
// keep(seed, idx) is a pure function of (seed, element index): forward and backward regenerate
// the same mask instead of storing it.  One murmur3-finaliser hash serves FOUR elements: with
// h1 = hash32(seed, idx >> 2) and h2 = xs15(h1 * 0x27D4EB2F), elements 4n .. 4n+3 take the 16-bit
// fields h1.lo, h1.hi, h2.lo, h2.hi and are kept when their field is >= thresh = round(p * 65536)
// (p resolved to 1.5e-5).  The fused epilogues and the attention kernels are VALU-bound and the hash's
// 32-bit multiplies run at quarter rate: four multiplies per four elements instead of six.  The
// derived fields were checked against the two-field form (tools/debug/hash_eval.py and the battery in
// tests/test_kernels_gpu.py: keep rate, serial correlation at tensor strides up to 2^28 elements, and the
// six pairings inside a quad, all within noise); two-round hashes on 24-bit multiplies, which would be
// cheaper still, were not (correlations of 4-50 sigma at power-of-two lags).
// Kernels whose lanes hold a 4-aligned run of elements use polus_keep4 (one hash per quad), the rest
// polus_keep -- same mask either way.  The per-call seed already mixes step / layer / site on the host.
__device__ __forceinline__ uint32_t polus_hash32(uint32_t seed, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t polus_hash32_second(uint32_t h1) {
    uint32_t y = h1 * 0x27D4EB2Fu;
    return y ^ (y >> 15);
}
__device__ __forceinline__ bool polus_keep(uint32_t seed, uint32_t idx, uint32_t thresh) {
    const uint32_t h1 = polus_hash32(seed, idx >> 2);
    const uint32_t h = (idx & 2u) ? polus_hash32_second(h1) : h1;
    return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thresh;
}
// idx4 must be a multiple of 4: masks of elements idx4 .. idx4 + 3
__device__ __forceinline__ void polus_keep4(uint32_t seed, uint32_t idx4, uint32_t thresh, bool (&k)[4]) {
    const uint32_t h1 = polus_hash32(seed, idx4 >> 2), h2 = polus_hash32_second(h1);
    k[0] = (h1 & 0xFFFFu) >= thresh;
    k[1] = (h1 >> 16) >= thresh;
    k[2] = (h2 & 0xFFFFu) >= thresh;
    k[3] = (h2 >> 16) >= thresh;
}
// v[r] = keep(base + r) ? v[r] * inv : 0 for r < N (N a multiple of 4); `quad` says base is a multiple of 4 (uniform)
template <int N>
__device__ __forceinline__ void polus_dropout_run(float (&v)[N], uint32_t seed, uint32_t base, uint32_t thresh, float inv, bool quad) {
    static_assert(N % 4 == 0, "runs of whole quads");
    if (quad) {
#pragma unroll
        for (int r = 0; r < N; r += 4) {
            bool k[4];
            polus_keep4(seed, base + r, thresh, k);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[r + e] = k[e] ? v[r + e] * inv : 0.f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = polus_keep(seed, base + r, thresh) ? v[r] * inv : 0.f;
    }
}
static inline uint32_t polus_drop_thresh(float p) {
    double t = (double)p * 65536.0 + 0.5;
    return t <= 0.0 ? 0u : (t >= 65535.0 ? 65535u : (uint32_t)t);
}

// ---------------------------------------------------------------- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- MFMA fragments
// A "fragment" is the 8 contraction elements k = 8*g + j (j = 0..7) that lane
// (i = lane & 15, g = lane >> 4) holds for tile row/column i.  bf16: one
// v_mfma_f32_16x16x32_bf16; f32: eight v_mfma_f32_16x16x4_f32, issue j taking element j
// (its k' = g maps to k = 8g + j, so all 32 k are covered once).  The f32 form is an
// exact k-ordered fmaf chain: deterministic and bit-stable.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

// D[row = 4g + r][col = i] += sum_k A[row][k] * B[k][col]; `a` supplies rows, `b` columns.
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Row-read fragment: 8 consecutive k of one tile row, K contiguous at `p` (16-B aligned).
__device__ __forceinline__ void frag_load_row(Frag<bf16_t>& f, const unsigned char* p) {
    f.v = *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void frag_load_row(Frag<float>& f, const unsigned char* p) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 16);
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

// Transposed-read of 4 k-rows x 16 columns of bf16 (ds_read_b64_tr_b16).  Lane i of each
// 16-lane group passes the address of (k-row i>>2, columns 4*(i&3)..+3) and receives
// column i of the four k-rows.  EXEC must be all ones.
__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}
