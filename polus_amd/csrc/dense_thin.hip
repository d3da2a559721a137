// Dense layers with a handful of output units (a token-classification head: hidden -> 4 labels; polus/ner/models.py:26-44 ends in
// one, tutorials/classifier_example.py:44-48 likewise): y = x W^T + b with W [C][H], C <= 8.
//
// A 128 x 128 MFMA tile wastes 97 % of its columns on such a layer, and its weight gradient is a [C][H] matrix contracted over
// every token -- three GEMM launches, a split-K reduce and two column-sum launches (141 us per step at 16384 tokens) for what is
// three passes over the activations.  Here: HBM-bound, a wave per row.
//   forward   reads x once, writes [rows][C]                                  (algorithmic bytes: rows * H * elt)
//   backward  reads x and dy once, writes dx, and keeps the C x H weight-gradient (and C bias-gradient) sums of its rows in
//             registers: workgroup partials in fixed wave order, summed over the workgroups in fixed order by one small launch
//             (deterministic)                                                  (rows * H * elt read + rows * H * elt written)
// Lane l holds VEC = 16 / sizeof(T) consecutive features of each 64 * VEC wide chunk, and W's fragment for them in registers.
#include "common.h"

namespace {

constexpr int TW = 4;                 // waves per workgroup
constexpr int THIN_MAX_BLOCKS = 512;

template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> { static constexpr int N = 8; };
template <> struct Vec16<float> { static constexpr int N = 4; };

template <typename T, int N> __device__ __forceinline__ void load_vec(const T* p, float (&v)[N]);
template <> __device__ __forceinline__ void load_vec<bf16_t, 8>(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
template <> __device__ __forceinline__ void load_vec<float, 4>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <typename T, int N> __device__ __forceinline__ void store_vec(T* p, const float (&v)[N]);
template <> __device__ __forceinline__ void store_vec<bf16_t, 8>(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
}
template <> __device__ __forceinline__ void store_vec<float, 4>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

// W fragment of this lane: w[c][j][e] = W[c][(j * 64 + lane) * VEC + e], zero beyond C or H
template <typename T, int CP, int NCH>
__device__ __forceinline__ void load_w(const T* W, long ldw, int C, int H, int lane, float (&w)[CP][NCH][Vec16<T>::N]) {
    constexpr int VEC = Vec16<T>::N;
#pragma unroll
    for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * VEC;
            if (c < C && col < H) load_vec<T, VEC>(W + (long)c * ldw + col, w[c][j]);
            else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) w[c][j][e] = 0.f;
            }
        }
}

template <typename T, typename TO, int CP, int NCH>
__global__ __launch_bounds__(64 * TW) void thin_fwd_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ W, long ldw,
                                                           const float* __restrict__ bias, TO* __restrict__ y, long ldy,
                                                           int rows, int H, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float w[CP][NCH][VEC];
    load_w<T, CP, NCH>(W, ldw, C, H, lane, w);
    float b = 0.f;
    if (bias && lane < C) b = bias[lane];
    for (int row = blockIdx.x * TW + wid; row < rows; row += gridDim.x * TW) {
        float s[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) s[c] = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * VEC;
            if (col < H) {
                float xv[VEC];
                load_vec<T, VEC>(x + (long)row * ldx + col, xv);
#pragma unroll
                for (int c = 0; c < CP; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) s[c] += xv[e] * w[c][j][e];
            }
        }
        float mine = 0.f;                  // lane c keeps the sum of unit c
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const float t = wave_sum(s[c]);
            if (lane == c) mine = t;
        }
        if (lane < C) y[(long)row * ldy + lane] = from_f<TO>(mine + b);
    }
}

// dy: [rows][C] in TD.  dx (may be null) [rows][H] in T.  partial: [gridDim.x][(C + 1)][Hp] floats, Hp = NCH * 64 * VEC:
// rows 0..C-1 the weight-gradient sums, row C holds the C bias-gradient sums in its first C entries.
template <typename T, typename TD, int CP, int NCH>
__global__ __launch_bounds__(64 * TW) void thin_bwd_kernel(const T* __restrict__ x, long ldx, const TD* __restrict__ dy, long lddy,
                                                           const T* __restrict__ W, long ldw, T* __restrict__ dx, long lddx,
                                                           float* __restrict__ partial, int rows, int H, int C) {
    constexpr int VEC = Vec16<T>::N, HP = NCH * 64 * VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);       // [CP][HP] weight-gradient sums + [CP] bias-gradient sums
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float w[CP][NCH][VEC], acc[CP][NCH][VEC], accb[CP];
    load_w<T, CP, NCH>(W, ldw, C, H, lane, w);
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        accb[c] = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[c][j][e] = 0.f;
    }
    for (int row = blockIdx.x * TW + wid; row < rows; row += gridDim.x * TW) {
        float d[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            d[c] = c < C ? to_f<TD>(dy[(long)row * lddy + c]) : 0.f;      // one address per wave: a broadcast load
            accb[c] += d[c];
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * VEC;
            if (col < H) {
                float xv[VEC], o[VEC];
                load_vec<T, VEC>(x + (long)row * ldx + col, xv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = 0.f;
#pragma unroll
                for (int c = 0; c < CP; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        o[e] += d[c] * w[c][j][e];
                        acc[c][j][e] += d[c] * xv[e];
                    }
                if (dx) store_vec<T, VEC>(dx + (long)row * lddx + col, o);
            }
        }
    }
    // the workgroup's sums: wave 0 lays its sums down, waves 1.. add theirs in turn (fixed order, one [CP][HP] image)
#pragma unroll
    for (int k = 0; k < TW; ++k) {
        if (wid == k) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float* q = lds + c * HP + (j * 64 + lane) * VEC + e;
                        *q = k == 0 ? acc[c][j][e] : *q + acc[c][j][e];
                    }
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    float* q = lds + CP * HP + c;
                    *q = k == 0 ? accb[c] : *q + accb[c];
                }
            }
        }
        __syncthreads();
    }
    float* dst = partial + (long)blockIdx.x * (C + 1) * HP;
    for (int idx = threadIdx.x; idx < C * HP; idx += blockDim.x) dst[idx] = lds[idx];
    if (threadIdx.x < C) dst[(long)C * HP + threadIdx.x] = lds[CP * HP + threadIdx.x];
}

// dW[c][h] (+)= sum_b partial[b][c][h], db[c] (+)= sum_b partial[b][C][c].  A workgroup owns 64 consecutive entries of the
// [(C + 1) * HP] partial row; its 16 thread groups take the blocks b = g, g + 16, ... and are combined in fixed order.
__global__ __launch_bounds__(1024) void thin_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int C, int H, int HP,
                                                                 float* __restrict__ dW, long lddw, float* __restrict__ db, int accumulate) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const long stride = (long)(C + 1) * HP;
    const int c = col / HP, h = col - c * HP;
    const bool live = c < C ? h < H : (c == C && h < C);
    float s = 0.f;
    if (live) {
#pragma unroll 8
        for (int b = gy; b < blocks; b += 16) s += partial[b * stride + col];
    }
    red[gy][cx] = s;
    __syncthreads();
    if (gy == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        if (c < C) {
            float* o = dW + (long)c * lddw + h;
            *o = accumulate ? *o + t : t;
        } else if (db) {
            db[h] = accumulate ? db[h] + t : t;
        }
    }
}

int thin_blocks(int rows) {
    int b = (rows + 2 * TW - 1) / (2 * TW);              // at least two rows per wave
    return b < 1 ? 1 : (b > THIN_MAX_BLOCKS ? THIN_MAX_BLOCKS : b);
}
int thin_nch(int dtype, int H) { const int span = 64 * (dtype == POLUS_BF16 ? 8 : 4); return (H + span - 1) / span; }
// chunks of the instantiation THIN_DISPATCH picks (bf16: 1 | 2, f32: 2 | 4): the row pitch of the partial sums
int thin_nch_inst(int dtype, int H) {
    const int n = thin_nch(dtype, H);
    return dtype == POLUS_BF16 ? (n <= 1 ? 1 : 2) : (n <= 2 ? 2 : 4);
}

}  // namespace

extern "C" int polus_dense_thin_supported(int dtype, int H, int C) {
    if (dtype != POLUS_BF16 && dtype != POLUS_F32) return 0;
    const int vec = dtype == POLUS_BF16 ? 8 : 4;
    const int nch = thin_nch(dtype, H);
    return C >= 1 && C <= 8 && H >= vec && H % vec == 0 && nch <= (dtype == POLUS_BF16 ? 2 : 4);
}

extern "C" size_t polus_dense_thin_bwd_workspace_bytes(int dtype, int rows, int H, int C) {
    const int hp = thin_nch_inst(dtype, H) * 64 * (dtype == POLUS_BF16 ? 8 : 4);
    return (size_t)thin_blocks(rows) * (C + 1) * hp * sizeof(float);
}

#define THIN_DISPATCH(CALL)                                                                          \
    do {                                                                                             \
        if (dtype == POLUS_BF16) {                                                                   \
            if (C <= 4) { if (nch <= 1) { CALL(bf16_t, 4, 1); } else { CALL(bf16_t, 4, 2); } }       \
            else        { if (nch <= 1) { CALL(bf16_t, 8, 1); } else { CALL(bf16_t, 8, 2); } }       \
        } else {                                                                                     \
            if (C <= 4) { if (nch <= 2) { CALL(float, 4, 2); } else { CALL(float, 4, 4); } }         \
            else        { if (nch <= 2) { CALL(float, 8, 2); } else { CALL(float, 8, 4); } }         \
        }                                                                                            \
    } while (0)

extern "C" int polus_dense_thin_fwd(int dtype, const void* x, long ldx, const void* W, long ldw, const float* bias,
                                    int y_dtype, void* y, long ldy, int rows, int H, int C, void* stream) {
    POLUS_REQUIRE(x && W && y && rows > 0, "polus_dense_thin_fwd: bad arguments");
    POLUS_REQUIRE(polus_dense_thin_supported(dtype, H, C), "polus_dense_thin_fwd: unsupported shape H=%d C=%d", H, C);
    POLUS_REQUIRE(y_dtype == POLUS_F32 || y_dtype == dtype, "polus_dense_thin_fwd: y_dtype must be f32 or dtype");
    const size_t es = polus_dtype_size(dtype);
    POLUS_REQUIRE(polus_aligned16(x) && polus_aligned16(W) && (ldx * es) % 16 == 0 && (ldw * es) % 16 == 0,
                  "polus_dense_thin_fwd: x and W rows must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nch = thin_nch(dtype, H), blocks = thin_blocks(rows);
#define THIN_FWD(T, CP, NCH)                                                                                                  \
    do {                                                                                                                      \
        if (y_dtype == POLUS_F32)                                                                                             \
            hipLaunchKernelGGL((thin_fwd_kernel<T, float, CP, NCH>), dim3(blocks), dim3(64 * TW), 0, st, (const T*)x, ldx,    \
                               (const T*)W, ldw, bias, (float*)y, ldy, rows, H, C);                                           \
        else                                                                                                                  \
            hipLaunchKernelGGL((thin_fwd_kernel<T, T, CP, NCH>), dim3(blocks), dim3(64 * TW), 0, st, (const T*)x, ldx,        \
                               (const T*)W, ldw, bias, (T*)y, ldy, rows, H, C);                                               \
    } while (0)
    THIN_DISPATCH(THIN_FWD);
#undef THIN_FWD
    POLUS_CHECK_LAUNCH("polus_dense_thin_fwd");
    return POLUS_OK;
}

extern "C" int polus_dense_thin_bwd(int dtype, const void* x, long ldx, int dy_dtype, const void* dy, long lddy,
                                    const void* W, long ldw, void* dx, long lddx, float* dW, long lddw, float* db,
                                    int rows, int H, int C, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(x && dy && W && dW && rows > 0, "polus_dense_thin_bwd: bad arguments");
    POLUS_REQUIRE(polus_dense_thin_supported(dtype, H, C), "polus_dense_thin_bwd: unsupported shape H=%d C=%d", H, C);
    POLUS_REQUIRE(dy_dtype == POLUS_F32 || dy_dtype == dtype, "polus_dense_thin_bwd: dy_dtype must be f32 or dtype");
    const size_t es = polus_dtype_size(dtype);
    POLUS_REQUIRE(polus_aligned16(x) && polus_aligned16(W) && (ldx * es) % 16 == 0 && (ldw * es) % 16 == 0 &&
                  (!dx || (polus_aligned16(dx) && (lddx * es) % 16 == 0)),
                  "polus_dense_thin_bwd: x, W and dx rows must be 16-byte aligned");
    const size_t need = polus_dense_thin_bwd_workspace_bytes(dtype, rows, H, C);
    if (!workspace || workspace_bytes < need) { polus_set_error("polus_dense_thin_bwd: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nch = thin_nch(dtype, H), blocks = thin_blocks(rows);
    const int hp = thin_nch_inst(dtype, H) * 64 * (dtype == POLUS_BF16 ? 8 : 4);
    float* partial = static_cast<float*>(workspace);
#define THIN_BWD(T, CP, NCH)                                                                                                   \
    do {                                                                                                                       \
        const size_t lds = ((size_t)CP * (NCH * 64 * Vec16<T>::N) + CP) * sizeof(float);                                         \
        if (dy_dtype == POLUS_F32)                                                                                             \
            hipLaunchKernelGGL((thin_bwd_kernel<T, float, CP, NCH>), dim3(blocks), dim3(64 * TW), lds, st, (const T*)x, ldx,   \
                               (const float*)dy, lddy, (const T*)W, ldw, (T*)dx, lddx, partial, rows, H, C);                   \
        else                                                                                                                   \
            hipLaunchKernelGGL((thin_bwd_kernel<T, T, CP, NCH>), dim3(blocks), dim3(64 * TW), lds, st, (const T*)x, ldx,       \
                               (const T*)dy, lddy, (const T*)W, ldw, (T*)dx, lddx, partial, rows, H, C);                       \
    } while (0)
    THIN_DISPATCH(THIN_BWD);
#undef THIN_BWD
    POLUS_CHECK_LAUNCH("polus_dense_thin_bwd");
    const int total = (C + 1) * hp;
    hipLaunchKernelGGL(thin_bwd_finalize_kernel, dim3((total + 63) / 64), dim3(1024), 0, st, partial, blocks, C, H, hp, dW, lddw, db,
                       accumulate ? 1 : 0);
    POLUS_CHECK_LAUNCH("polus_dense_thin_bwd(finalize)");
    return POLUS_OK;
}
