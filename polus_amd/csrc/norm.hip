// LayerNorm fwd/bwd, embedding gather+LN fwd/bwd and deterministic column sums (gfx950).
// All HBM-bound: one wave64 owns one row, lanes read 4 consecutive features per chunk
// (8-B bf16 / 16-B f32 accesses, 512 B / 1 KiB per wave-instruction), row statistics by
// wave reductions, cross-row (per-feature) sums accumulated in registers over a
// grid-stride row loop and finished by an order-fixed two-stage reduction (no atomics,
// bitwise reproducible).
#include "common.h"

namespace {

constexpr int MAXC = 8;           // chunks of 256 features per row: H <= 2048 (template NC <= MAXC)
constexpr int WAVES = 16;         // waves per workgroup (1024 threads): 4096 waves at 256 workgroups
constexpr int LN_THREADS = 64 * WAVES;
constexpr int MAX_PARTIAL_BLOCKS = 256;   // per-feature partial sums [blocks][3][H] f32, flushed once per 64 rows

__device__ __forceinline__ int n_chunks(int H) { return (H + 255) >> 8; }

template <typename T, int NC>
__device__ __forceinline__ void load_row(const T* row, int H, int lane, float (&v)[NC][4]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) load4<T>(row + col, v[c]);
        else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
    }
}

// per-feature f32 vector (gamma / beta) -> registers, once per wave
template <int NC>
__device__ __forceinline__ void load_feat(const float* __restrict__ p, int H, int lane, float (&v)[NC][4]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) load4<float>(p + col, v[c]);
        else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
    }
}

// mean / rstd of one row held in registers (two-pass, biased variance)
template <int NC>
__device__ __forceinline__ void row_stats(const float (&v)[NC][4], int H, int lane, float eps,
                                          float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[c][e] - mean; q += d * d; }
        }
    }
    float var = wave_sum(q) / (float)H;
    rstd = 1.0f / sqrtf(var + eps);
}

template <typename T, int NC>
__device__ __forceinline__ void normalize_store(const float (&v)[NC][4], const float (&gv)[NC][4], const float (&bv)[NC][4],
                                                T* y, int H, int lane, float mean, float rstd,
                                                unsigned dthresh = 0, unsigned dseed = 0, float dinv = 1.f, unsigned rowbase = 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * gv[c][e] + bv[c][e];
            // rowbase = row * H and col are multiples of 4: even-aligned run
            if (dthresh) polus_dropout_run<4>(o, dseed, rowbase + col, dthresh, dinv, true);
            store4<T>(y + col, o);
        }
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int rows, int H, float eps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float gv[NC][4], bv[NC][4];
    load_feat<NC>(gamma, H, lane, gv);
    load_feat<NC>(beta, H, lane, bv);
    for (int row = blockIdx.x * WAVES + wid; row < rows; row += gridDim.x * WAVES) {
        float v[NC][4];
        load_row<T, NC>(x + (long)row * H, H, lane, v);
        float mu, rs;
        row_stats<NC>(v, H, lane, eps, mu, rs);
        normalize_store<T, NC>(v, gv, bv, y + (long)row * H, H, lane, mu, rs);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// ---- bf16, H % 256 == 0: a HALF-wave per row, 16-byte accesses.  Lane (half = lane >> 5, hl = lane & 31) owns
// columns 8 (hl + 32 c) .. +7 of row 2 w + half: one wave-instruction moves 2 x 512 contiguous bytes (two rows) at
// 16 B per lane instead of 512 B at 8 B per lane -- 8-byte accesses run at 0.54-0.70 of the 16-byte rate.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int NC>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_hw_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                int rows, int H, float eps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
    float gv[NC][8], bv[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (hl + 32 * c) * 8;
        load4<float>(gamma + col, *reinterpret_cast<float(*)[4]>(&gv[c][0])); load4<float>(gamma + col + 4, *reinterpret_cast<float(*)[4]>(&gv[c][4]));
        load4<float>(beta + col, *reinterpret_cast<float(*)[4]>(&bv[c][0])); load4<float>(beta + col + 4, *reinterpret_cast<float(*)[4]>(&bv[c][4]));
    }
    const float invH = 1.0f / (float)H;
    const int nw = blockDim.x >> 6;                      // waves of this workgroup (the launch picks 4 or 16)
    for (int row = (blockIdx.x * nw + wid) * 2 + half; row < rows; row += gridDim.x * nw * 2) {
        float v[NC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bf16x8v t = *reinterpret_cast<const bf16x8v*>(x + (long)row * H + (hl + 32 * c) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s += v[c][e]; }
        }
        const float mu = half_sum(s) * invH;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mu; q += d * d; }
        const float rs = 1.0f / sqrtf(half_sum(q) * invH + eps);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            bf16x8v o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((v[c][e] - mu) * rs * gv[c][e] + bv[c][e]);
            *reinterpret_cast<bf16x8v*>(y + (long)row * H + (hl + 32 * c) * 8) = o;
        }
        if (hl == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

struct DropArgs { unsigned thresh, seed; float inv; const PolusDyn* dyn = nullptr; };   // thresh == 0: no dropout; dyn: see common.h

// Shared tail of the LN backward kernels: given x-hat pieces and dy for one row, produce dx
// and accumulate the per-feature sums.
template <int NC> struct ColAcc { float dg[NC][4], db[NC][4], dbias[NC][4]; };

template <int NC>
__device__ __forceinline__ void colacc_zero(ColAcc<NC>& a) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) a.dg[c][e] = a.db[c][e] = a.dbias[c][e] = 0.f;
}

// block-level, order-fixed combine of the 4 waves' column accumulators into
// partial[block][3][H]
template <int NC>
__device__ __forceinline__ void colacc_flush(const ColAcc<NC>& a, float* lds /*[3*H]*/, float* partial, int H,
                                             int lane, int wid, int want_bias) {
    for (int w = 0; w < WAVES; ++w) {
        if (wid == w) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                int col = (lane + 64 * c) * 4;
                if (col < H) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (w == 0) {
                            lds[col + e] = a.dg[c][e]; lds[H + col + e] = a.db[c][e]; lds[2 * H + col + e] = a.dbias[c][e];
                        } else {
                            lds[col + e] += a.dg[c][e]; lds[H + col + e] += a.db[c][e]; lds[2 * H + col + e] += a.dbias[c][e];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    float* dst = partial + (long)blockIdx.x * 3 * H;
    int n = (want_bias ? 3 : 2) * H;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) dst[idx] = lds[idx];
}

// Same for small workgroups (W waves): every wave drops its sums into its own LDS slice
// [W][3H], then all threads add the W slices in fixed order -- one barrier instead of W.
template <int NC, int W>
__device__ __forceinline__ void colacc_flush_par(const ColAcc<NC>& a, float* lds /*[W][3*H]*/, float* partial, int H,
                                                 int lane, int wid, int want_bias) {
    float* mine = lds + (long)wid * 3 * H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) {
            *reinterpret_cast<float4*>(mine + col) = make_float4(a.dg[c][0], a.dg[c][1], a.dg[c][2], a.dg[c][3]);
            *reinterpret_cast<float4*>(mine + H + col) = make_float4(a.db[c][0], a.db[c][1], a.db[c][2], a.db[c][3]);
            *reinterpret_cast<float4*>(mine + 2 * H + col) = make_float4(a.dbias[c][0], a.dbias[c][1], a.dbias[c][2], a.dbias[c][3]);
        }
    }
    __syncthreads();
    float* dst = partial + (long)blockIdx.x * 3 * H;
    int n = (want_bias ? 3 : 2) * H;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        float t = lds[idx];
#pragma unroll
        for (int w = 1; w < W; ++w) t += lds[(long)w * 3 * H + idx];
        dst[idx] = t;
    }
}

template <typename T, typename TDX, int NC>
__device__ __forceinline__ void ln_bwd_row(const float (&xv)[NC][4], const T* dyrow, const float (&gv)[NC][4],
                                           TDX* dxrow, int H, int lane, float mu, float rs, ColAcc<NC>& acc,
                                           int want_bias, TDX* dxm_row = nullptr, DropArgs out_drop = DropArgs{0, 0, 1.f},
                                           DropArgs in_drop = DropArgs{0, 0, 1.f}, unsigned rowbase = 0) {
    float dy[NC][4];
    load_row<T, NC>(dyrow, H, lane, dy);
    if (in_drop.thresh) {   // y = dropout(LN(x)): the incoming gradient passes through the same mask
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            int col = (lane + 64 * c) * 4;
            polus_dropout_run<4>(dy[c], in_drop.seed, rowbase + col, in_drop.thresh, in_drop.inv, true);
        }
    }
    float s1 = 0.f, s2 = 0.f;
    float xh[NC][4], dxh[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[c][e] = (xv[c][e] - mu) * rs;
                dxh[c][e] = dy[c][e] * gv[c][e];
                s1 += dxh[c][e];
                s2 += dxh[c][e] * xh[c][e];
                acc.dg[c][e] += dy[c][e] * xh[c][e];
                acc.db[c][e] += dy[c][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[c][e] = dxh[c][e] = 0.f;
        }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) {
            float o[4], om[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (dxh[c][e] - s1 - xh[c][e] * s2) * rs; om[e] = o[e]; }
            // x = dropout(dense) + residual: the Dense (and its bias) see the masked gradient
            if (out_drop.thresh) polus_dropout_run<4>(om, out_drop.seed, rowbase + col, out_drop.thresh, out_drop.inv, true);
            if (want_bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc.dbias[c][e] += om[e];
            }
            store4<TDX>(dxrow + col, o);
            if (dxm_row) store4<TDX>(dxm_row + col, om);
        }
    }
}

// LayerNorm backward proper runs 4-wave workgroups.  At ~210 VGPRs two of them fit a CU, so 512 are resident at
// once: the default cap (POLUS_LN_BWD_BLOCKS).  Their [blocks][3H] partials are reduced in fixed order by one
// finalize launch up to POLUS_LN_FIN_SINGLE (512) rows, above that in two stages, groups of FIN_GROUP first.
// Single-stream kernel time per step (profiles/r02_ln_bwd_reduce_shapes.txt): 1024 blocks / two stages 25.8 us +
// 2 x 4.8 us per LayerNorm; 512 / two stages 22.0 + 2 x 4.7; 512 / one stage 21.8 + 4.8.
constexpr int BWD_WAVES = 4, BWD_MAX_BLOCKS = 1024, FIN_GROUP = 128;
template <typename T, int NC>
__global__ __launch_bounds__(64 * BWD_WAVES) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     float* __restrict__ partial, int rows, int H, int want_bias,
                                                     T* __restrict__ dxm, DropArgs drop) {
    if (drop.thresh) drop.seed = polus_eff_seed(drop.seed, drop.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    ColAcc<NC> acc;
    colacc_zero(acc);
    float gv[NC][4];
    load_feat<NC>(gamma, H, lane, gv);
    for (int row = blockIdx.x * BWD_WAVES + wid; row < rows; row += gridDim.x * BWD_WAVES) {
        float xv[NC][4];
        load_row<T, NC>(x + (long)row * H, H, lane, xv);
        ln_bwd_row<T, T, NC>(xv, dy + (long)row * H, gv, dx + (long)row * H, H, lane, mean[row], rstd[row], acc, want_bias,
                             dxm ? dxm + (long)row * H : nullptr, drop, DropArgs{0, 0, 1.f}, (unsigned)row * (unsigned)H);
    }
    colacc_flush_par<NC, BWD_WAVES>(acc, lds, partial, H, lane, wid, want_bias);
}

// out_k[c] (+)= sum_p partial[p][k*seg + c]: 64 columns x 16 partial groups per block,
// groups combined in fixed order.
// blockIdx.y selects a group of `pgroup` consecutive partial rows (first stage of a two-stage
// reduction: out0 then is a [groups][ncols] array, seg = ncols, written at row blockIdx.y).
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partial, int P, int pstride,
                                                               int ncols, int seg, float* out0, float* out1,
                                                               float* out2, int accumulate, int pgroup = 0) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (pgroup > 0) {
        const int p0 = blockIdx.y * pgroup;
        partial += (long)p0 * pstride;
        P = min(pgroup, P - p0);
        out0 += (long)blockIdx.y * ncols;
    }
    if (col < ncols) {
        // fixed order p = gy, gy + 16, ...; eight loads in flight (the rows are latency-, not bandwidth-bound)
#pragma unroll 8
        for (int p = gy; p < P; p += 16) s += partial[(long)p * pstride + col];
    }
    red[gy][cx] = s;
    __syncthreads();
    if (gy == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        int which = col / seg, c = col % seg;
        float* out = which == 0 ? out0 : which == 1 ? out1 : out2;
        if (out) out[c] = accumulate ? out[c] + t : t;
    }
}

// LayerNorm backward, half-wave per row (see ln_fwd_hw_kernel).  The two half-waves of a wave own the SAME columns
// (of two different rows), so the per-feature sums of a wave are the lane-wise sums of its halves (one xor-32
// shuffle at flush time); the block's waves are then combined through LDS exactly as in the wave-per-row kernel.
template <int NC>
__global__ __launch_bounds__(64 * BWD_WAVES) void ln_bwd_hw_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                                    float* __restrict__ partial, int rows, int H, int want_bias,
                                                                    bf16_t* __restrict__ dxm, DropArgs drop) {
    if (drop.thresh) drop.seed = polus_eff_seed(drop.seed, drop.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
    float gv[NC][8], adg[NC][8], adb[NC][8], abias[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (hl + 32 * c) * 8;
        load4<float>(gamma + col, *reinterpret_cast<float(*)[4]>(&gv[c][0])); load4<float>(gamma + col + 4, *reinterpret_cast<float(*)[4]>(&gv[c][4]));
#pragma unroll
        for (int e = 0; e < 8; ++e) adg[c][e] = adb[c][e] = abias[c][e] = 0.f;
    }
    const float invH = 1.0f / (float)H;
    for (int row = (blockIdx.x * BWD_WAVES + wid) * 2 + half; row < rows; row += gridDim.x * BWD_WAVES * 2) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NC][8], dxh[NC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const long off = (long)row * H + (hl + 32 * c) * 8;
            const bf16x8v tx = *reinterpret_cast<const bf16x8v*>(x + off);
            const bf16x8v td = *reinterpret_cast<const bf16x8v*>(dy + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)td[e];
                xh[c][e] = ((float)tx[e] - mu) * rs;
                dxh[c][e] = d * gv[c][e];
                s1 += dxh[c][e];
                s2 += dxh[c][e] * xh[c][e];
                adg[c][e] += d * xh[c][e];
                adb[c][e] += d;
            }
        }
        s1 = half_sum(s1) * invH;
        s2 = half_sum(s2) * invH;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = (hl + 32 * c) * 8;
            float o[8], om[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = (dxh[c][e] - s1 - xh[c][e] * s2) * rs; om[e] = o[e]; }
            if (drop.thresh) polus_dropout_run<8>(om, drop.seed, (unsigned)row * (unsigned)H + col, drop.thresh, drop.inv, true);
            if (want_bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) abias[c][e] += om[e];
            }
            bf16x8v t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (bf16_t)o[e];
            *reinterpret_cast<bf16x8v*>(dx + (long)row * H + col) = t;
            if (dxm) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (bf16_t)om[e];
                *reinterpret_cast<bf16x8v*>(dxm + (long)row * H + col) = t;
            }
        }
    }
    // the wave's sums = its two halves, lane-wise; then the block's waves through LDS in fixed order
    float* mine = lds + (long)wid * 3 * H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (hl + 32 * c) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            adg[c][e] += __shfl_xor(adg[c][e], 32, 64);
            adb[c][e] += __shfl_xor(adb[c][e], 32, 64);
            abias[c][e] += __shfl_xor(abias[c][e], 32, 64);
        }
        if (half == 0) {
            *reinterpret_cast<float4*>(mine + col) = make_float4(adg[c][0], adg[c][1], adg[c][2], adg[c][3]);
            *reinterpret_cast<float4*>(mine + col + 4) = make_float4(adg[c][4], adg[c][5], adg[c][6], adg[c][7]);
            *reinterpret_cast<float4*>(mine + H + col) = make_float4(adb[c][0], adb[c][1], adb[c][2], adb[c][3]);
            *reinterpret_cast<float4*>(mine + H + col + 4) = make_float4(adb[c][4], adb[c][5], adb[c][6], adb[c][7]);
            *reinterpret_cast<float4*>(mine + 2 * H + col) = make_float4(abias[c][0], abias[c][1], abias[c][2], abias[c][3]);
            *reinterpret_cast<float4*>(mine + 2 * H + col + 4) = make_float4(abias[c][4], abias[c][5], abias[c][6], abias[c][7]);
        }
    }
    __syncthreads();
    float* dst = partial + (long)blockIdx.x * 3 * H;
    const int n = (want_bias ? 3 : 2) * H;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        float t = lds[idx];
#pragma unroll
        for (int w = 1; w < BWD_WAVES; ++w) t += lds[(long)w * 3 * H + idx];
        dst[idx] = t;
    }
}

// ---- generic column sums: grid (col tiles of 1024, row chunks); thread owns 4 columns
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long ldx, int rows, int cols,
                                                             int rows_per_chunk, float* __restrict__ partial,
                                                             const int32_t* __restrict__ sel, int sel_value, int vec) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < cols) {
        const int nv = cols - col;
        for (int r = r0; r < r1; ++r) {
            if (sel && sel[r] != sel_value) continue;
            float v[4];
            const T* p = x + (long)r * ldx + col;
            if (vec && nv >= 4) load4<T>(p, v);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < nv ? to_f<T>(p[e]) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nv) partial[(long)blockIdx.y * cols + col + e] = a[e];
    }
}

int colsum_launch(int dtype, const void* x, long ldx, int rows, int cols, float* out, int accumulate,
                  const int32_t* sel, int sel_value, void* workspace, size_t workspace_bytes, hipStream_t st) {
    int chunks = (rows + 63) / 64;
    if (chunks > 256) chunks = 256;
    int rpc = (rows + chunks - 1) / chunks;
    chunks = (rows + rpc - 1) / rpc;
    size_t need = (size_t)chunks * cols * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        polus_set_error("polus_colsum: workspace %zu < %zu", workspace_bytes, need);
        return POLUS_ERR_WORKSPACE;
    }
    float* partial = static_cast<float*>(workspace);
    dim3 grid((cols + 1023) / 1024, chunks);
    size_t es = polus_dtype_size(dtype);
    int vec = (((uintptr_t)x) % (4 * es) == 0) && (ldx % 4 == 0);
    if (dtype == POLUS_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, rows, cols, rpc, partial, sel, sel_value, vec);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, rows, cols, rpc, partial, sel, sel_value, vec);
    POLUS_CHECK_LAUNCH("polus_colsum(partial)");
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((cols + 63) / 64), dim3(1024), 0, st,
                       partial, chunks, cols, cols, cols, out, (float*)nullptr, (float*)nullptr, accumulate);
    POLUS_CHECK_LAUNCH("polus_colsum(finalize)");
    return POLUS_OK;
}

// pick the smallest instantiated chunk count covering H
#define POLUS_NC_DISPATCH(H, T, KERNEL, ...)                                                   \
    do {                                                                                       \
        int nc__ = ((H) + 255) / 256;                                                          \
        if (nc__ <= 1) hipLaunchKernelGGL((KERNEL<T, 1>), __VA_ARGS__);                        \
        else if (nc__ <= 2) hipLaunchKernelGGL((KERNEL<T, 2>), __VA_ARGS__);                   \
        else if (nc__ <= 3) hipLaunchKernelGGL((KERNEL<T, 3>), __VA_ARGS__);                   \
        else if (nc__ <= 4) hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<T, 8>), __VA_ARGS__);                                  \
    } while (0)

int ln_bwd_blocks(int rows) {
    int b = (rows + BWD_WAVES - 1) / BWD_WAVES;
    int cap = polus_cfg().ln_bwd_blocks;
    cap = cap < 64 ? 64 : (cap > BWD_MAX_BLOCKS ? BWD_MAX_BLOCKS : cap);
    return b > cap ? cap : (b < 1 ? 1 : b);
}
int ln_blocks(int rows) {
    int b = (rows + WAVES - 1) / WAVES;
    return b > MAX_PARTIAL_BLOCKS ? MAX_PARTIAL_BLOCKS : (b < 1 ? 1 : b);
}

// ---------------------------------------------------------------- embeddings
template <typename T, int NC>
__device__ __forceinline__ void gather_sum(const float* word, const float* pos, const float* type, int id, int s,
                                           int tt, int H, int lane, float (&v)[NC][4]) {
    const float* w = word + (long)id * H;
    const float* p = pos + (long)s * H;
    const float* t = type + (long)tt * H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int col = (lane + 64 * c) * 4;
        if (col < H) {
            float a[4], b[4], d[4];
            load4<float>(w + col, a); load4<float>(p + col, b); load4<float>(t + col, d);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[c][e] = (a[e] + d[e]) + b[e];  // word + type + pos (oracle order)
        } else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(LN_THREADS) void embed_fwd_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ tts,
                                                        const float* __restrict__ word, const float* __restrict__ pos,
                                                        const float* __restrict__ type, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        int B, int S, int H, int vocab, int type_vocab, float eps,
                                                        unsigned dthresh, unsigned dseed, float dinv, const PolusDyn* dyn) {
    if (dthresh) dseed = polus_eff_seed(dseed, dyn);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rows = B * S;
    float gv[NC][4], bv[NC][4];
    load_feat<NC>(gamma, H, lane, gv);
    load_feat<NC>(beta, H, lane, bv);
    for (int row = blockIdx.x * WAVES + wid; row < rows; row += gridDim.x * WAVES) {
        int id = ids[row]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        int tt = tts ? tts[row] : 0; tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
        float v[NC][4];
        gather_sum<T, NC>(word, pos, type, id, row % S, tt, H, lane, v);
        float mu, rs;
        row_stats<NC>(v, H, lane, eps, mu, rs);
        normalize_store<T, NC>(v, gv, bv, y + (long)row * H, H, lane, mu, rs, dthresh, dseed, dinv, (unsigned)row * (unsigned)H);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// LN backward of the embedding sum: de (f32 workspace) + gamma/beta partials
template <typename T, int NC>
__global__ __launch_bounds__(LN_THREADS) void embed_bwd_ln_kernel(const T* __restrict__ dy, const int32_t* __restrict__ ids,
                                                           const int32_t* __restrict__ tts, const float* __restrict__ word,
                                                           const float* __restrict__ pos, const float* __restrict__ type,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ de,
                                                           float* __restrict__ partial, int B, int S, int H, int vocab,
                                                           int type_vocab, DropArgs in_drop) {
    if (in_drop.thresh) in_drop.seed = polus_eff_seed(in_drop.seed, in_drop.dyn);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rows = B * S;
    ColAcc<NC> acc;
    colacc_zero(acc);
    float gv[NC][4];
    load_feat<NC>(gamma, H, lane, gv);
    for (int row = blockIdx.x * WAVES + wid; row < rows; row += gridDim.x * WAVES) {
        int id = ids[row]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        int tt = tts ? tts[row] : 0; tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
        float xv[NC][4];
        gather_sum<T, NC>(word, pos, type, id, row % S, tt, H, lane, xv);
        ln_bwd_row<T, float, NC>(xv, dy + (long)row * H, gv, de + (long)row * H, H, lane, mean[row], rstd[row], acc, 0,
                                 nullptr, DropArgs{0, 0, 1.f}, in_drop, (unsigned)row * (unsigned)H);
    }
    colacc_flush(acc, lds, partial, H, lane, wid, 0);
}

// word-table gradient, atomic form.  f32 atomics run at the memory side at ~1.3 TB/s when spread over rows but 14x
// slower when many adders meet on ONE row (MI355X_MICROARCH.md, Global float atomics), and a quarter of a padded
// batch is the [PAD] id, in runs at the end of every sequence.  Each wave therefore takes SC_RUN consecutive tokens,
// loads all of their rows first (every row is needed exactly once; vmcnt retires in order, so a load issued after
// an atomic would wait for it), combines the duplicates among them in registers -- the first occurrence sums, in
// token order -- and issues one atomic row-add per distinct id, 256 contiguous bytes per wave-instruction.  No LDS,
// no workgroup synchronisation, no wave sums more than SC_RUN rows.  168 -> 79 us with the combining alone at the
// headline shape (runs of 4: 96 us, of 16: 140 us -- fewer, longer waves).  NC = ceil(H / 256) <= 4.
constexpr int SC_RUN = 8, SC_WAVES = 8;
template <int NC>
__global__ __launch_bounds__(64 * SC_WAVES) void embed_scatter_atomic_kernel(const float* __restrict__ de, const int32_t* __restrict__ ids,
                                                                      float* __restrict__ gword, int rows, int H, int vocab) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * SC_WAVES * SC_RUN;
    for (long r0 = ((long)blockIdx.x * SC_WAVES + wid) * SC_RUN; r0 < rows; r0 += stride) {
        int mine = -1;                                       // lanes 0 .. SC_RUN-1 hold the (clamped) ids of the run
        if (lane < SC_RUN && r0 + lane < rows) { mine = ids[r0 + lane]; mine = mine < 0 ? 0 : (mine >= vocab ? vocab - 1 : mine); }
        float row[SC_RUN][NC][4];
#pragma unroll
        for (int j = 0; j < SC_RUN; ++j) {
            const bool valid = r0 + j < rows;                // wave-uniform
            const float* src = de + (r0 + j) * H + lane;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    row[j][c][k] = (valid && c * 256 + k * 64 + lane < H) ? src[c * 256 + k * 64] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < SC_RUN; ++t) {
            const int id = __shfl(mine, t, 64);              // wave-uniform
            const unsigned same = (unsigned)__ballot(mine == id);
            if (id < 0 || (same & ((1u << t) - 1u))) continue;   // past the last row, or an earlier token owns this id
            float* dst = gword + (long)id * H + lane;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float acc = row[t][c][k];
#pragma unroll
                    for (int j = t + 1; j < SC_RUN; ++j)
                        if ((same >> j) & 1u) acc += row[j][c][k];
                    if (c * 256 + k * 64 + lane < H) atomicAdd(dst + c * 256 + k * 64, acc);
                }
        }
    }
}
// any H: one wave per token, no combining
__global__ __launch_bounds__(LN_THREADS) void embed_scatter_atomic_wide_kernel(const float* __restrict__ de, const int32_t* __restrict__ ids,
                                                                        float* __restrict__ gword, int rows, int H, int vocab) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int row = blockIdx.x * WAVES + wid; row < rows; row += gridDim.x * WAVES) {
        int id = ids[row]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float* src = de + (long)row * H;
        float* dst = gword + (long)id * H;
        for (int col = lane; col < H; col += 64) atomicAdd(dst + col, src[col]);
    }
}

// word-table gradient, reproducible form: the first occurrence of an id owns it and adds
// the rows of every occurrence in token order.
__global__ __launch_bounds__(LN_THREADS) void embed_scatter_owner_kernel(const float* __restrict__ de, const int32_t* __restrict__ ids,
                                                                  float* __restrict__ gword, int rows, int H, int vocab) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int row = blockIdx.x * WAVES + wid; row < rows; row += gridDim.x * WAVES) {
        // duplicates are found on the CLAMPED id: two different out-of-range ids land on the same table row
        auto clampid = [vocab](int v) { return v < 0 ? 0 : (v >= vocab ? vocab - 1 : v); };
        const int cid = clampid(ids[row]);
        bool dup = false;
        for (int j0 = 0; j0 < row && !dup; j0 += 64) {
            int j = j0 + lane;
            bool hit = (j < row) && (clampid(ids[j]) == cid);
            dup = __any(hit);
        }
        if (dup) continue;  // wave-uniform
        float* dst = gword + (long)cid * H;
        for (int c0 = 0; c0 < H; c0 += 64 * 4) {  // 256-feature slabs held in registers
            int col = c0 + lane * 4;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j0 = row; j0 < rows; j0 += 64) {
                int j = j0 + lane;
                unsigned long long m = __ballot((j < rows) && (clampid(ids[j]) == cid));
                while (m) {
                    int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (col < H) {
                        float v[4];
                        load4<float>(de + (long)(j0 + b) * H + col, v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] += v[e];
                    }
                }
            }
            if (col < H) {
                float o[4];
                load4<float>(dst + col, o);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += a[e];
                store4<float>(dst + col, o);
            }
        }
    }
}

// position-table gradient: gpos[s] (+)= sum_b de[b, s]  (fixed b order)
__global__ __launch_bounds__(256) void embed_pos_grad_kernel(const float* __restrict__ de, float* __restrict__ gpos,
                                                             int B, int S, int H, int accumulate) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)S * H) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += de[(long)b * S * H + idx];
    gpos[idx] = accumulate ? gpos[idx] + s : s;
}

}  // namespace

extern "C" size_t polus_layernorm_bwd_workspace_bytes(int rows, int H) {
    // [blocks][3H] block partials (max of the two users: LayerNorm proper, embedding LayerNorm)
    // + [groups][3H] second-stage partials
    int b = ln_bwd_blocks(rows), b2 = ln_blocks(rows);
    if (b2 > b) b = b2;
    int groups = (b + FIN_GROUP - 1) / FIN_GROUP;
    return ((size_t)b + groups) * 3 * (size_t)H * sizeof(float);
}

extern "C" int polus_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta,
                                   void* y, float* mean, float* rstd, int rows, int H, float eps, void* stream) {
    POLUS_REQUIRE(x && gamma && beta && y && mean && rstd, "polus_layernorm_fwd: null pointer");
    POLUS_REQUIRE(rows > 0 && H > 0 && H % 4 == 0 && H <= 256 * MAXC, "polus_layernorm_fwd: H=%d must be a multiple of 4, <= %d", H, 256 * MAXC);
    POLUS_REQUIRE(polus_aligned16(x) && polus_aligned16(y) && polus_aligned16(gamma) && polus_aligned16(beta),
                  "polus_layernorm_fwd: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = (rows + WAVES - 1) / WAVES;
    if (blocks > 4096) blocks = 4096;
    if (dtype == POLUS_BF16 && H % 256 == 0 && H <= 1024 && rows % 2 == 0 && polus_cfg().ln_halfwave) {
        // 4-wave workgroups: a 16-wave one (98 registers) fills a CU alone, so the launch ran as two rounds of 256 workgroups that all
        // load, then all store; small workgroups keep five per CU in different phases (10.4 -> 9.4 us at 16384 x 768, tools/ln_bench.py, round 4)
        const int fw = 4;
        int hb = (rows / 2 + fw - 1) / fw;
        if (hb > 16384) hb = 16384;
        const int LN_THREADS_HW = 64 * fw;
        if (H == 256) hipLaunchKernelGGL(ln_fwd_hw_kernel<1>, dim3(hb), dim3(LN_THREADS_HW), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, H, eps);
        else if (H == 512) hipLaunchKernelGGL(ln_fwd_hw_kernel<2>, dim3(hb), dim3(LN_THREADS_HW), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, H, eps);
        else if (H == 768) hipLaunchKernelGGL(ln_fwd_hw_kernel<3>, dim3(hb), dim3(LN_THREADS_HW), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, H, eps);
        else hipLaunchKernelGGL(ln_fwd_hw_kernel<4>, dim3(hb), dim3(LN_THREADS_HW), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, H, eps);
    } else if (dtype == POLUS_BF16)
        POLUS_NC_DISPATCH(H, bf16_t, ln_fwd_kernel, dim3(blocks), dim3(LN_THREADS), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, H, eps);
    else if (dtype == POLUS_F32)
        POLUS_NC_DISPATCH(H, float, ln_fwd_kernel, dim3(blocks), dim3(LN_THREADS), 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, rows, H, eps);
    else POLUS_FAIL("polus_layernorm_fwd: bad dtype");
    POLUS_CHECK_LAUNCH("polus_layernorm_fwd");
    return POLUS_OK;
}

extern "C" int polus_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                                   const float* mean, const float* rstd, void* dx,
                                   float* dgamma, float* dbeta, float* dbias, int accumulate,
                                   int rows, int H, void* dx_masked, float drop_p, uint32_t seed,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    // dgamma == dbeta == null: leave the per-workgroup partial sums in `workspace` (which the caller then owns until
    // polus_layernorm_bwd_finalize has run on it, on any stream ordered behind this call); `dbias` non-null still says
    // that the bias-gradient column sums are wanted
    const bool defer = !dgamma && !dbeta;
    POLUS_REQUIRE(dy && x && gamma && mean && rstd && dx && (defer || (dgamma && dbeta)), "polus_layernorm_bwd: null pointer");
    POLUS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (long)rows * H < (1LL << 32), "polus_layernorm_bwd: bad dropout arguments");
    POLUS_REQUIRE(!(drop_p > 0.f) || dx_masked, "polus_layernorm_bwd: dropout needs dx_masked");
    DropArgs drop{drop_p > 0.f ? polus_drop_thresh(drop_p) : 0u, seed, 1.0f / (1.0f - drop_p), polus_dyn()};
    POLUS_REQUIRE(rows > 0 && H > 0 && H % 4 == 0 && H <= 256 * MAXC, "polus_layernorm_bwd: bad H=%d", H);
    POLUS_REQUIRE(polus_aligned16(x) && polus_aligned16(dy) && polus_aligned16(dx) && polus_aligned16(gamma),
                  "polus_layernorm_bwd: pointers must be 16-byte aligned");
    size_t need = polus_layernorm_bwd_workspace_bytes(rows, H);
    if (!workspace || workspace_bytes < need) { polus_set_error("polus_layernorm_bwd: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = ln_bwd_blocks(rows);
    float* partial = static_cast<float*>(workspace);
    size_t lds = (size_t)BWD_WAVES * 3 * (size_t)H * sizeof(float);
    int wb = dbias ? 1 : 0;
    if (dtype == POLUS_BF16 && H % 256 == 0 && H <= 1024 && rows % 2 == 0 && polus_cfg().ln_halfwave && polus_aligned16(dx_masked)) {
#define POLUS_LN_BWD_HW(NCV) hipLaunchKernelGGL(ln_bwd_hw_kernel<NCV>, dim3(blocks), dim3(64 * BWD_WAVES), lds, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx, partial, rows, H, wb, (bf16_t*)dx_masked, drop)
        if (H == 256) POLUS_LN_BWD_HW(1); else if (H == 512) POLUS_LN_BWD_HW(2); else if (H == 768) POLUS_LN_BWD_HW(3); else POLUS_LN_BWD_HW(4);
#undef POLUS_LN_BWD_HW
    } else if (dtype == POLUS_BF16)
        POLUS_NC_DISPATCH(H, bf16_t, ln_bwd_kernel, dim3(blocks), dim3(64 * BWD_WAVES), lds, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx, partial, rows, H, wb, (bf16_t*)dx_masked, drop);
    else if (dtype == POLUS_F32)
        POLUS_NC_DISPATCH(H, float, ln_bwd_kernel, dim3(blocks), dim3(64 * BWD_WAVES), lds, st, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, partial, rows, H, wb, (float*)dx_masked, drop);
    else POLUS_FAIL("polus_layernorm_bwd: bad dtype");
    POLUS_CHECK_LAUNCH("polus_layernorm_bwd");
    if (defer) return POLUS_OK;
    int ncols = (wb ? 3 : 2) * H;
    if (blocks > polus_cfg().ln_fin_single) {
        // two fixed-order stages: [blocks] -> [groups] -> result (a single stage would leave most
        // of the chip idle: ncols/64 workgroups walking 1024 rows each)
        int groups = (blocks + FIN_GROUP - 1) / FIN_GROUP;
        float* part2 = partial + (size_t)blocks * 3 * H;
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64, groups), dim3(1024), 0, st,
                           partial, blocks, 3 * H, ncols, ncols, part2, (float*)nullptr, (float*)nullptr, 0, FIN_GROUP);
        POLUS_CHECK_LAUNCH("polus_layernorm_bwd(finalize 1)");
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, st,
                           part2, groups, ncols, ncols, H, dgamma, dbeta, dbias, accumulate, 0);
    } else {
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, st,
                           partial, blocks, 3 * H, ncols, H, dgamma, dbeta, dbias, accumulate, 0);
    }
    POLUS_CHECK_LAUNCH("polus_layernorm_bwd(finalize)");
    return POLUS_OK;
}

// Second half of polus_layernorm_bwd when it was called with dgamma = dbeta = null: the fixed-order reduction of the
// [workgroups][3H] partials left in `workspace` into dgamma / dbeta (/ dbias).  Same kernels, same order, same result.
extern "C" int polus_layernorm_bwd_finalize(void* workspace, size_t workspace_bytes, int rows, int H, float* dgamma, float* dbeta,
                                            float* dbias, int accumulate, void* stream) {
    POLUS_REQUIRE(workspace && dgamma && dbeta && rows > 0 && H > 0 && H % 4 == 0, "polus_layernorm_bwd_finalize: bad arguments");
    size_t need = polus_layernorm_bwd_workspace_bytes(rows, H);
    if (workspace_bytes < need) { polus_set_error("polus_layernorm_bwd_finalize: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blocks = ln_bwd_blocks(rows);
    float* partial = static_cast<float*>(workspace);
    const int ncols = (dbias ? 3 : 2) * H;
    if (blocks > polus_cfg().ln_fin_single) {
        int groups = (blocks + FIN_GROUP - 1) / FIN_GROUP;
        float* part2 = partial + (size_t)blocks * 3 * H;
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64, groups), dim3(1024), 0, st,
                           partial, blocks, 3 * H, ncols, ncols, part2, (float*)nullptr, (float*)nullptr, 0, FIN_GROUP);
        POLUS_CHECK_LAUNCH("polus_layernorm_bwd_finalize(1)");
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, st,
                           part2, groups, ncols, ncols, H, dgamma, dbeta, dbias, accumulate, 0);
    } else {
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, st,
                           partial, blocks, 3 * H, ncols, H, dgamma, dbeta, dbias, accumulate, 0);
    }
    POLUS_CHECK_LAUNCH("polus_layernorm_bwd_finalize");
    return POLUS_OK;
}

extern "C" size_t polus_colsum_workspace_bytes(int rows, int cols) {
    int chunks = (rows + 63) / 64;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    return (size_t)chunks * (size_t)cols * sizeof(float);
}

extern "C" int polus_colsum(int dtype, const void* x, long ldx, int rows, int cols, float* out, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(x && out && rows > 0 && cols > 0 && ldx >= cols, "polus_colsum: bad arguments");
    POLUS_REQUIRE(dtype == POLUS_F32 || dtype == POLUS_BF16, "polus_colsum: bad dtype");
    return colsum_launch(dtype, x, ldx, rows, cols, out, accumulate, nullptr, 0, workspace, workspace_bytes,
                         static_cast<hipStream_t>(stream));
}

extern "C" size_t polus_embed_bwd_workspace_bytes(int B, int S, int H) {
    size_t rows = (size_t)B * S;
    size_t de = rows * H * sizeof(float);
    size_t part = polus_layernorm_bwd_workspace_bytes((int)rows, H);
    size_t cs = polus_colsum_workspace_bytes((int)rows, H);
    return de + (part > cs ? part : cs) + 256;
}

extern "C" int polus_embed_ln_fwd(int dtype, const int32_t* ids, const int32_t* type_ids,
                                  const float* word, const float* pos, const float* type,
                                  const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                  int B, int S, int H, int vocab, int max_pos, int type_vocab, float eps,
                                  float drop_p, uint32_t seed, void* stream) {
    POLUS_REQUIRE(ids && word && pos && type && gamma && beta && y && mean && rstd, "polus_embed_ln_fwd: null pointer");
    POLUS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "polus_embed_ln_fwd: bad drop_p");
    const unsigned dthresh = drop_p > 0.f ? polus_drop_thresh(drop_p) : 0u;
    const float dinv = 1.0f / (1.0f - drop_p);
    POLUS_REQUIRE(B > 0 && S > 0 && S <= max_pos, "polus_embed_ln_fwd: S=%d exceeds max_position_embeddings=%d", S, max_pos);
    POLUS_REQUIRE(H % 4 == 0 && H <= 256 * MAXC && vocab > 0 && type_vocab > 0, "polus_embed_ln_fwd: bad H=%d", H);
    POLUS_REQUIRE(polus_aligned16(word) && polus_aligned16(pos) && polus_aligned16(type) && polus_aligned16(y) &&
                  polus_aligned16(gamma) && polus_aligned16(beta), "polus_embed_ln_fwd: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rows = B * S, blocks = (rows + WAVES - 1) / WAVES;
    if (blocks > 4096) blocks = 4096;
    if (dtype == POLUS_BF16)
        POLUS_NC_DISPATCH(H, bf16_t, embed_fwd_kernel, dim3(blocks), dim3(LN_THREADS), 0, st, ids, type_ids, word, pos, type, gamma, beta, (bf16_t*)y, mean, rstd, B, S, H, vocab, type_vocab, eps, dthresh, seed, dinv, polus_dyn());
    else if (dtype == POLUS_F32)
        POLUS_NC_DISPATCH(H, float, embed_fwd_kernel, dim3(blocks), dim3(LN_THREADS), 0, st, ids, type_ids, word, pos, type, gamma, beta, (float*)y, mean, rstd, B, S, H, vocab, type_vocab, eps, dthresh, seed, dinv, polus_dyn());
    else POLUS_FAIL("polus_embed_ln_fwd: bad dtype");
    POLUS_CHECK_LAUNCH("polus_embed_ln_fwd");
    return POLUS_OK;
}

extern "C" int polus_embed_ln_bwd(int dtype, const void* dy, const int32_t* ids, const int32_t* type_ids,
                                  const float* word, const float* pos, const float* type, const float* gamma,
                                  const float* mean, const float* rstd,
                                  float* gword, float* gpos, float* gtype, float* ggamma, float* gbeta,
                                  int accumulate, int deterministic,
                                  int B, int S, int H, int vocab, int max_pos, int type_vocab,
                                  float drop_p, uint32_t seed,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    DropArgs in_drop{drop_p > 0.f ? polus_drop_thresh(drop_p) : 0u, seed, 1.0f / (1.0f - drop_p), polus_dyn()};
    POLUS_REQUIRE(dy && ids && word && pos && type && gamma && mean && rstd && gword && gpos && gtype && ggamma && gbeta,
                  "polus_embed_ln_bwd: null pointer");
    POLUS_REQUIRE(B > 0 && S > 0 && S <= max_pos && H % 4 == 0 && H <= 256 * MAXC, "polus_embed_ln_bwd: bad shape");
    POLUS_REQUIRE(polus_aligned16(dy) && polus_aligned16(gword) && polus_aligned16(workspace),
                  "polus_embed_ln_bwd: pointers must be 16-byte aligned");
    size_t need = polus_embed_bwd_workspace_bytes(B, S, H);
    if (!workspace || workspace_bytes < need) { polus_set_error("polus_embed_ln_bwd: workspace %zu < %zu", workspace_bytes, need); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = B * S;
    float* de = static_cast<float*>(workspace);
    size_t de_bytes = ((size_t)rows * H * sizeof(float) + 255) / 256 * 256;
    float* partial = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + de_bytes);
    size_t partial_bytes = workspace_bytes - de_bytes;
    int blocks = ln_blocks(rows);
    size_t lds = 3 * (size_t)H * sizeof(float);
    if (dtype == POLUS_BF16)
        POLUS_NC_DISPATCH(H, bf16_t, embed_bwd_ln_kernel, dim3(blocks), dim3(LN_THREADS), lds, st, (const bf16_t*)dy, ids, type_ids, word, pos, type, gamma, mean, rstd, de, partial, B, S, H, vocab, type_vocab, in_drop);
    else if (dtype == POLUS_F32)
        POLUS_NC_DISPATCH(H, float, embed_bwd_ln_kernel, dim3(blocks), dim3(LN_THREADS), lds, st, (const float*)dy, ids, type_ids, word, pos, type, gamma, mean, rstd, de, partial, B, S, H, vocab, type_vocab, in_drop);
    else POLUS_FAIL("polus_embed_ln_bwd: bad dtype");
    POLUS_CHECK_LAUNCH("polus_embed_ln_bwd(ln)");
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((2 * H + 63) / 64), dim3(1024), 0, st,
                       partial, blocks, 3 * H, 2 * H, H, ggamma, gbeta, (float*)nullptr, accumulate);
    POLUS_CHECK_LAUNCH("polus_embed_ln_bwd(finalize)");

    if (!accumulate) {
        POLUS_HIP(hipMemsetAsync(gword, 0, (size_t)vocab * H * sizeof(float), st));
        POLUS_HIP(hipMemsetAsync(gpos, 0, (size_t)max_pos * H * sizeof(float), st));
    }
    int sblocks = (rows + WAVES - 1) / WAVES;
    if (sblocks > 4096) sblocks = 4096;
    if (deterministic)
        hipLaunchKernelGGL(embed_scatter_owner_kernel, dim3(sblocks), dim3(LN_THREADS), 0, st, de, ids, gword, rows, H, vocab);
    else
    {
        const dim3 g(min((rows + SC_RUN * SC_WAVES - 1) / (SC_RUN * SC_WAVES), 4096)), b(64 * SC_WAVES);
        if (H <= 256) hipLaunchKernelGGL(embed_scatter_atomic_kernel<1>, g, b, 0, st, de, ids, gword, rows, H, vocab);
        else if (H <= 512) hipLaunchKernelGGL(embed_scatter_atomic_kernel<2>, g, b, 0, st, de, ids, gword, rows, H, vocab);
        else if (H <= 768) hipLaunchKernelGGL(embed_scatter_atomic_kernel<3>, g, b, 0, st, de, ids, gword, rows, H, vocab);
        else if (H <= 1024) hipLaunchKernelGGL(embed_scatter_atomic_kernel<4>, g, b, 0, st, de, ids, gword, rows, H, vocab);
        else hipLaunchKernelGGL(embed_scatter_atomic_wide_kernel, dim3(sblocks), dim3(LN_THREADS), 0, st, de, ids, gword, rows, H, vocab);
    }
    POLUS_CHECK_LAUNCH("polus_embed_ln_bwd(scatter)");
    hipLaunchKernelGGL(embed_pos_grad_kernel, dim3(((long)S * H + 255) / 256), dim3(256), 0, st, de, gpos, B, S, H, accumulate);
    POLUS_CHECK_LAUNCH("polus_embed_ln_bwd(pos)");
    for (int t = 0; t < type_vocab; ++t) {
        if (type_ids) {
            int rc = colsum_launch(POLUS_F32, de, H, rows, H, gtype + (long)t * H, accumulate, type_ids, t, partial, partial_bytes, st);
            if (rc != POLUS_OK) return rc;
        } else if (t == 0) {
            int rc = colsum_launch(POLUS_F32, de, H, rows, H, gtype, accumulate, nullptr, 0, partial, partial_bytes, st);
            if (rc != POLUS_OK) return rc;
        } else if (!accumulate) {
            POLUS_HIP(hipMemsetAsync(gtype + (long)t * H, 0, (size_t)H * sizeof(float), st));
        }
    }
    return POLUS_OK;
}
