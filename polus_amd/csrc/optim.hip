// Fused multi-tensor Adam / AdamWeightDecay over the flat f32 parameter arena (gfx950).
// HBM-bound: 7 f32 streams per parameter (read p,g,m,v; write p,m,v = 28 B) plus the
// optional 2-B bf16 shadow write of GEMM weights; 16-B accesses, grid-stride.
// Update rule: Keras Adam as HF AdamWeightDecay applies it (see oracle/optim.py).
#include "common.h"

namespace {

constexpr int SEG_CHUNK_ELEMS = 1 << 14;  // host tables cut tensors into <= 16K-element chunks

struct AdamArgs {
    float* p; const float* g; float* m; float* v; bf16_t* shadow;
    const int64_t* seg; int n_seg;
    float lr, lr_t, b1, b2, eps, wd, gscale;
    const float* clip;
    const PolusDyn* dyn;   // graph replay: lr and lr_t of this step come from device memory
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    if (a.dyn) { a.lr = a.dyn->lr; a.lr_t = a.dyn->lr_t; }
    const float gs = a.gscale * (a.clip ? *a.clip : 1.0f);
    for (int s = blockIdx.x; s < a.n_seg; s += gridDim.x) {
        const int64_t beg = a.seg[3 * s], end = a.seg[3 * s + 1], flags = a.seg[3 * s + 2];
        const float decay = (flags & 1) ? a.lr * a.wd : 0.0f;
        const bool shadow = (flags & 2) && a.shadow;
        // head (unaligned) elements, then 16-byte body, then tail
        int64_t body = (beg + 3) & ~(int64_t)3;
        if (body > end) body = end;
        const int64_t body_end = body + ((end - body) & ~(int64_t)3);
        for (int64_t i = beg + threadIdx.x; i < body; i += 256) {
            float g = a.g[i] * gs, p = a.p[i], m = a.m[i], v = a.v[i];
            p -= decay * p;
            m = a.b1 * m + (1.0f - a.b1) * g;
            v = a.b2 * v + (1.0f - a.b2) * g * g;
            p -= a.lr_t * m / (sqrtf(v) + a.eps);
            a.p[i] = p; a.m[i] = m; a.v[i] = v;
            if (shadow) a.shadow[i] = (bf16_t)p;
        }
        for (int64_t i = body + 4 * (int64_t)threadIdx.x; i < body_end; i += 4 * 256) {
            float4 g4 = *reinterpret_cast<const float4*>(a.g + i);
            float4 p4 = *reinterpret_cast<float4*>(a.p + i);
            float4 m4 = *reinterpret_cast<float4*>(a.m + i);
            float4 v4 = *reinterpret_cast<float4*>(a.v + i);
            float g[4] = {g4.x * gs, g4.y * gs, g4.z * gs, g4.w * gs};
            float p[4] = {p4.x, p4.y, p4.z, p4.w}, m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p[e] -= decay * p[e];
                m[e] = a.b1 * m[e] + (1.0f - a.b1) * g[e];
                v[e] = a.b2 * v[e] + (1.0f - a.b2) * g[e] * g[e];
                p[e] -= a.lr_t * m[e] / (sqrtf(v[e]) + a.eps);
            }
            *reinterpret_cast<float4*>(a.p + i) = make_float4(p[0], p[1], p[2], p[3]);
            *reinterpret_cast<float4*>(a.m + i) = make_float4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<float4*>(a.v + i) = make_float4(v[0], v[1], v[2], v[3]);
            if (shadow) store4<bf16_t>(a.shadow + i, p);
        }
        for (int64_t i = body_end + threadIdx.x; i < end; i += 256) {
            float g = a.g[i] * gs, p = a.p[i], m = a.m[i], v = a.v[i];
            p -= decay * p;
            m = a.b1 * m + (1.0f - a.b1) * g;
            v = a.b2 * v + (1.0f - a.b2) * g * g;
            p -= a.lr_t * m / (sqrtf(v) + a.eps);
            a.p[i] = p; a.m[i] = m; a.v[i] = v;
            if (shadow) a.shadow[i] = (bf16_t)p;
        }
    }
}

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 v = *reinterpret_cast<const float4*>(g + i);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (int64_t j = i; j < n; ++j) s += g[j] * g[j];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// sum of squares over the windows [seg[3k], seg[3k+1]) of g: block b takes segments b, b + grid, ...
__global__ __launch_bounds__(256) void sqnorm_seg_kernel(const float* __restrict__ g, const int64_t* __restrict__ seg, int n_seg,
                                                         float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int k = blockIdx.x; k < n_seg; k += gridDim.x) {
        const int64_t b = seg[3 * k], e = seg[3 * k + 1];
        for (int64_t i = b + threadIdx.x; i < e; i += 256) s += g[i] * g[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void sqnorm_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) s += partial[k];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += red[k];
        *out = t;
    }
}

__global__ void clip_scale_kernel(const float* __restrict__ sq, int n, float gscale, float clip, float* __restrict__ out) {
    float t = 0.f;
    for (int k = 0; k < n; ++k) t += sq[k];
    float norm = sqrtf(t) * fabsf(gscale);
    *out = clip / fmaxf(norm, clip);  // tf.clip_by_global_norm: scale = clip / max(norm, clip)
}

int sq_blocks(int64_t n) {
    int64_t b = (n / 4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" int polus_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16,
                               const int64_t* seg, int n_seg, int64_t n,
                               float lr, float lr_t, float beta1, float beta2, float eps, float weight_decay,
                               float grad_scale, const float* clip_scale, void* stream) {
    POLUS_REQUIRE(p && g && m && v && seg && n_seg > 0 && n > 0, "polus_adam_step: bad arguments");
    POLUS_REQUIRE(polus_aligned16(p) && polus_aligned16(g) && polus_aligned16(m) && polus_aligned16(v),
                  "polus_adam_step: arenas must be 16-byte aligned");
    POLUS_REQUIRE(!shadow_bf16 || ((uintptr_t)shadow_bf16 % 8) == 0, "polus_adam_step: shadow must be 8-byte aligned");
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.shadow = static_cast<bf16_t*>(shadow_bf16);
    a.seg = seg; a.n_seg = n_seg;
    a.dyn = polus_dyn();
    a.lr = lr; a.lr_t = lr_t; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.gscale = grad_scale; a.clip = clip_scale;
    int blocks = n_seg < 4096 ? n_seg : 4096;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    POLUS_CHECK_LAUNCH("polus_adam_step");
    return POLUS_OK;
}

extern "C" size_t polus_sqnorm_workspace_bytes(int64_t n) { return (size_t)sq_blocks(n) * sizeof(float); }

extern "C" int polus_sqnorm(const float* g, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(g && out && n > 0 && polus_aligned16(g), "polus_sqnorm: bad arguments");
    if (!workspace || workspace_bytes < polus_sqnorm_workspace_bytes(n)) { polus_set_error("polus_sqnorm: workspace too small"); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = sq_blocks(n);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(256), 0, st, g, n, static_cast<float*>(workspace));
    POLUS_CHECK_LAUNCH("polus_sqnorm(partial)");
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, static_cast<const float*>(workspace), blocks, out);
    POLUS_CHECK_LAUNCH("polus_sqnorm(final)");
    return POLUS_OK;
}

extern "C" int polus_sqnorm_segments(const float* g, const int64_t* seg, int n_seg, float* out,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(g && seg && out && n_seg > 0, "polus_sqnorm_segments: bad arguments");
    const int blocks = n_seg < 1024 ? n_seg : 1024;
    if (!workspace || workspace_bytes < (size_t)1024 * sizeof(float)) { polus_set_error("polus_sqnorm_segments: workspace too small"); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sqnorm_seg_kernel, dim3(blocks), dim3(256), 0, st, g, seg, n_seg, static_cast<float*>(workspace));
    POLUS_CHECK_LAUNCH("polus_sqnorm_segments(partial)");
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, static_cast<const float*>(workspace), blocks, out);
    POLUS_CHECK_LAUNCH("polus_sqnorm_segments(final)");
    return POLUS_OK;
}

extern "C" int polus_clip_scale(const float* sqnorm, int n_terms, float grad_scale, float clip_norm, float* out_scale, void* stream) {
    POLUS_REQUIRE(sqnorm && out_scale && clip_norm > 0.f && n_terms > 0, "polus_clip_scale: bad arguments");
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), sqnorm, n_terms, grad_scale, clip_norm, out_scale);
    POLUS_CHECK_LAUNCH("polus_clip_scale");
    return POLUS_OK;
}
