// Losses of the Polus training step (gfx950): sparse / class-weighted softmax CE,
// class-weighted sigmoid CE, linear-chain CRF negative log-likelihood + Viterbi, argmax.
// All are tiny next to the encoder (C <= a few dozen classes): one thread per row (CE) or
// per sequence (CRF scan), f32 arithmetic, per-block partial losses combined in a fixed
// order by a second single-block kernel (bitwise reproducible, no atomics).
#include "common.h"

namespace {

constexpr int CRF_MAXC = 16;

// sum `partial[0..n)` in index order -> *out * mul
__global__ void finalize_sum_kernel(const float* __restrict__ partial, int n, float mul, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) s += partial[k];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += red[k];
        *out = t * mul;
    }
}

__device__ __forceinline__ void block_partial(float v, float* partial) {
    __shared__ float red[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits, long ldl,
                                                           const int32_t* __restrict__ labels,
                                                           const float* __restrict__ cw, float* __restrict__ partial,
                                                           T* __restrict__ dlogits, long lddl, int rows, int C) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (row < rows) {
        const float* x = logits + (long)row * ldl;
        int lab = labels[row];
        lab = lab < 0 ? 0 : (lab >= C ? C - 1 : lab);
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        const float w = cw ? cw[lab] : 1.0f;
        loss = (lse - x[lab]) * w;
        const float inv = w / (float)rows;
        T* d = dlogits + (long)row * lddl;
        for (int c = 0; c < C; ++c) {
            float pr = expf(x[c] - lse);
            d[c] = from_f<T>((pr - (c == lab ? 1.0f : 0.0f)) * inv);
        }
    }
    block_partial(loss, partial);
}

template <typename T>
__global__ __launch_bounds__(256) void sigmoid_xent_kernel(const float* __restrict__ logits, long ldl,
                                                           const float* __restrict__ y, long ldy,
                                                           const float* __restrict__ cw, float neg_w,
                                                           float* __restrict__ partial, T* __restrict__ dlogits,
                                                           long lddl, int rows, int C) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (row < rows) {
        const float* x = logits + (long)row * ldl;
        const float* yt = y + (long)row * ldy;
        bool allzero = true;
        float w = 0.f;
        for (int c = 0; c < C; ++c) { allzero = allzero && (yt[c] == 0.0f); w += cw[c] * yt[c]; }
        if (allzero) w += neg_w;
        float un = 0.f;
        for (int c = 0; c < C; ++c) un += fmaxf(x[c], 0.f) - x[c] * yt[c] + log1pf(expf(-fabsf(x[c])));
        loss = un * w;
        const float inv = w / (float)rows;
        T* d = dlogits + (long)row * lddl;
        for (int c = 0; c < C; ++c) d[c] = from_f<T>((1.0f / (1.0f + expf(-x[c])) - yt[c]) * inv);
    }
    block_partial(loss, partial);
}

__global__ void argmax_kernel(const float* __restrict__ x, long ldx, int32_t* __restrict__ out, int rows, int C) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const float* p = x + (long)row * ldx;
    int best = 0;
    float bv = p[0];
    for (int c = 1; c < C; ++c) if (p[c] > bv) { bv = p[c]; best = c; }  // first maximum, as tf.argmax
    out[row] = best;
}

__device__ __forceinline__ float lse_n(const float* v, int n) {
    float mx = v[0];
    for (int k = 1; k < n; ++k) mx = fmaxf(mx, v[k]);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(v[k] - mx);
    return mx + logf(s);
}

// One thread per sequence: forward (alpha) scan, path score, backward (beta) scan with
// marginals -> gradients.  alpha is kept in the workspace [B][S][C].
template <typename T>
__global__ void crf_nll_kernel(const float* __restrict__ pot, const int32_t* __restrict__ tags,
                               const int32_t* __restrict__ lengths, const float* __restrict__ trans,
                               const float* __restrict__ sw, float* __restrict__ nll_b, T* __restrict__ dpot,
                               float* __restrict__ dtrans_b, float* __restrict__ alpha_ws, int B, int S, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float Tm[CRF_MAXC * CRF_MAXC];
    for (int k = 0; k < C * C; ++k) Tm[k] = trans[k];
    const float* x = pot + (long)b * S * C;
    const int32_t* t = tags + (long)b * S;
    float* al = alpha_ws + (long)b * S * C;
    T* dx = dpot + (long)b * S * C;
    float* dT = dtrans_b + (long)b * C * C;
    for (int k = 0; k < C * C; ++k) dT[k] = 0.f;
    int L = lengths ? lengths[b] : S;
    L = L < 0 ? 0 : (L > S ? S : L);
    const float w = (sw ? sw[b] : 1.0f) / (float)B;  // d(mean(-ll*w))/d(ll) = -w/B
    for (int s = L; s < S; ++s) for (int c = 0; c < C; ++c) dx[(long)s * C + c] = from_f<T>(0.f);
    if (L == 0) { nll_b[b] = 0.f; return; }
    // path score
    float score = 0.f;
    for (int s = 0; s < L; ++s) {
        int ts = t[s]; ts = ts < 0 ? 0 : (ts >= C ? C - 1 : ts);
        score += x[(long)s * C + ts];
        if (s + 1 < L) { int tn = t[s + 1]; tn = tn < 0 ? 0 : (tn >= C ? C - 1 : tn); score += Tm[ts * C + tn]; }
    }
    // alpha
    float prev[CRF_MAXC], cur[CRF_MAXC], tmp[CRF_MAXC];
    for (int c = 0; c < C; ++c) { prev[c] = x[c]; al[c] = prev[c]; }
    for (int s = 1; s < L; ++s) {
        for (int j = 0; j < C; ++j) {
            for (int k = 0; k < C; ++k) tmp[k] = prev[k] + Tm[k * C + j];
            cur[j] = lse_n(tmp, C) + x[(long)s * C + j];
        }
        for (int c = 0; c < C; ++c) { prev[c] = cur[c]; al[(long)s * C + c] = cur[c]; }
    }
    const float logz = lse_n(prev, C);
    nll_b[b] = -(score - logz) * (sw ? sw[b] : 1.0f);
    // beta + gradients: d(-ll)/dx = marginal - onehot ; d(-ll)/dT = pair marginal - onehot pair
    float beta[CRF_MAXC], nb[CRF_MAXC];
    for (int c = 0; c < C; ++c) beta[c] = 0.f;
    for (int s = L - 1; s >= 0; --s) {
        int ts = t[s]; ts = ts < 0 ? 0 : (ts >= C ? C - 1 : ts);
        for (int c = 0; c < C; ++c) {
            float marg = expf(al[(long)s * C + c] + beta[c] - logz);
            dx[(long)s * C + c] = from_f<T>((marg - (c == ts ? 1.0f : 0.0f)) * w);
        }
        if (s > 0) {
            int tp = t[s - 1]; tp = tp < 0 ? 0 : (tp >= C ? C - 1 : tp);
            for (int k = 0; k < C; ++k) {
                for (int j = 0; j < C; ++j) {
                    float e = x[(long)s * C + j] + beta[j];
                    tmp[j] = Tm[k * C + j] + e;
                    float pair = expf(al[(long)(s - 1) * C + k] + tmp[j] - logz);
                    dT[k * C + j] += (pair - ((k == tp && j == ts) ? 1.0f : 0.0f)) * w;
                }
                nb[k] = lse_n(tmp, C);
            }
            for (int c = 0; c < C; ++c) beta[c] = nb[c];
        }
    }
}

// loss = mean_b nll_b ; dtrans (+)= sum_b dtrans_b  (fixed b order)
__global__ void crf_finalize_kernel(const float* __restrict__ nll_b, const float* __restrict__ dtrans_b, int B, int C,
                                    float* __restrict__ loss, float* __restrict__ dtrans, int accumulate) {
    int k = threadIdx.x;
    if (k < C * C) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dtrans_b[(long)b * C * C + k];
        dtrans[k] = accumulate ? dtrans[k] + s : s;
    }
    if (k == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += nll_b[b];
        *loss = s / (float)B;
    }
}

__global__ void crf_viterbi_kernel(const float* __restrict__ pot, const int32_t* __restrict__ lengths,
                                   const float* __restrict__ trans, int32_t* __restrict__ out,
                                   int32_t* __restrict__ back_ws, int B, int S, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* x = pot + (long)b * S * C;
    int32_t* back = back_ws + (long)b * S * C;
    int32_t* o = out + (long)b * S;
    int L = lengths ? lengths[b] : S;
    L = L < 0 ? 0 : (L > S ? S : L);
    for (int s = 0; s < S; ++s) o[s] = 0;
    if (L == 0) return;
    float score[CRF_MAXC], ns[CRF_MAXC];
    for (int c = 0; c < C; ++c) score[c] = x[c];
    for (int s = 1; s < L; ++s) {
        for (int j = 0; j < C; ++j) {
            int bi = 0;
            float bv = score[0] + trans[j];
            for (int k = 1; k < C; ++k) {
                float v = score[k] + trans[k * C + j];
                if (v > bv) { bv = v; bi = k; }
            }
            back[(long)s * C + j] = bi;
            ns[j] = bv + x[(long)s * C + j];
        }
        for (int c = 0; c < C; ++c) score[c] = ns[c];
    }
    int best = 0;
    for (int c = 1; c < C; ++c) if (score[c] > score[best]) best = c;
    o[L - 1] = best;
    for (int s = L - 1; s > 0; --s) { best = back[(long)s * C + best]; o[s - 1] = best; }
}

}  // namespace

extern "C" size_t polus_loss_workspace_bytes(int rows) { return ((size_t)(rows + 255) / 256 + 1) * sizeof(float); }

extern "C" int polus_softmax_xent(int dtype, const float* logits, long ldl, const int32_t* labels,
                                  const float* class_weights, float* loss, void* dlogits, long lddl,
                                  int rows, int C, void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(logits && labels && loss && dlogits, "polus_softmax_xent: null pointer");
    POLUS_REQUIRE(rows > 0 && C > 0 && ldl >= C && lddl >= C, "polus_softmax_xent: bad shape rows=%d C=%d", rows, C);
    if (!workspace || workspace_bytes < polus_loss_workspace_bytes(rows)) { polus_set_error("polus_softmax_xent: workspace too small"); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = (rows + 255) / 256;
    float* partial = static_cast<float*>(workspace);
    if (dtype == POLUS_BF16)
        hipLaunchKernelGGL(softmax_xent_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, logits, ldl, labels, class_weights, partial, (bf16_t*)dlogits, lddl, rows, C);
    else if (dtype == POLUS_F32)
        hipLaunchKernelGGL(softmax_xent_kernel<float>, dim3(blocks), dim3(256), 0, st, logits, ldl, labels, class_weights, partial, (float*)dlogits, lddl, rows, C);
    else POLUS_FAIL("polus_softmax_xent: bad dtype");
    POLUS_CHECK_LAUNCH("polus_softmax_xent");
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, st, partial, blocks, 1.0f / (float)rows, loss);
    POLUS_CHECK_LAUNCH("polus_softmax_xent(finalize)");
    return POLUS_OK;
}

extern "C" int polus_sigmoid_xent(int dtype, const float* logits, long ldl, const float* y_true, long ldy,
                                  const float* class_weights, float negative_weight, float* loss,
                                  void* dlogits, long lddl, int rows, int C,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(logits && y_true && class_weights && loss && dlogits, "polus_sigmoid_xent: null pointer");
    POLUS_REQUIRE(rows > 0 && C > 0 && ldl >= C && lddl >= C && ldy >= C, "polus_sigmoid_xent: bad shape");
    if (!workspace || workspace_bytes < polus_loss_workspace_bytes(rows)) { polus_set_error("polus_sigmoid_xent: workspace too small"); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = (rows + 255) / 256;
    float* partial = static_cast<float*>(workspace);
    if (dtype == POLUS_BF16)
        hipLaunchKernelGGL(sigmoid_xent_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, logits, ldl, y_true, ldy, class_weights, negative_weight, partial, (bf16_t*)dlogits, lddl, rows, C);
    else if (dtype == POLUS_F32)
        hipLaunchKernelGGL(sigmoid_xent_kernel<float>, dim3(blocks), dim3(256), 0, st, logits, ldl, y_true, ldy, class_weights, negative_weight, partial, (float*)dlogits, lddl, rows, C);
    else POLUS_FAIL("polus_sigmoid_xent: bad dtype");
    POLUS_CHECK_LAUNCH("polus_sigmoid_xent");
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, st, partial, blocks, 1.0f / (float)rows, loss);
    POLUS_CHECK_LAUNCH("polus_sigmoid_xent(finalize)");
    return POLUS_OK;
}

// cm[r[i]][c[i]] += 1 for i < n (int32, exact): per-workgroup histogram in LDS, one atomic per non-zero cell
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ r, const int32_t* __restrict__ c,
                                                        int64_t n, int C, int32_t* __restrict__ cm,
                                                        int32_t* __restrict__ rejected) {
    extern __shared__ int hist[];
    const int cells = C * C;
    for (int k = threadIdx.x; k < cells; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int a = r[i], b = c[i];
        if ((unsigned)a < (unsigned)C && (unsigned)b < (unsigned)C) atomicAdd(&hist[a * C + b], 1);
        else if (rejected) atomicAdd(rejected, 1);      // out of [0, C): counted, never silently dropped
    }
    __syncthreads();
    for (int k = threadIdx.x; k < cells; k += blockDim.x)
        if (hist[k]) atomicAdd(&cm[k], hist[k]);
}

extern "C" int polus_confusion_matrix(const int32_t* row_idx, const int32_t* col_idx, int64_t n, int C,
                                      int32_t* cm, int32_t* rejected, void* stream) {
    POLUS_REQUIRE(row_idx && col_idx && cm && n >= 0 && C > 0 && C <= 128, "polus_confusion_matrix: bad arguments (0 < C <= 128)");
    if (n == 0) return POLUS_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(confusion_kernel, dim3(blocks), dim3(256), (size_t)C * C * sizeof(int), static_cast<hipStream_t>(stream),
                       row_idx, col_idx, n, C, cm, rejected);
    POLUS_CHECK_LAUNCH("polus_confusion_matrix");
    return POLUS_OK;
}

extern "C" int polus_argmax(const float* x, long ldx, int32_t* out, int rows, int C, void* stream) {
    POLUS_REQUIRE(x && out && rows > 0 && C > 0 && ldx >= C, "polus_argmax: bad arguments");
    hipLaunchKernelGGL(argmax_kernel, dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, out, rows, C);
    POLUS_CHECK_LAUNCH("polus_argmax");
    return POLUS_OK;
}

extern "C" size_t polus_crf_workspace_bytes(int B, int S, int C) {
    // alpha (or viterbi back-pointers) [B,S,C] + per-sequence nll [B] + per-sequence dtrans [B,C,C]
    return ((size_t)B * S * C + (size_t)B + (size_t)B * C * C) * sizeof(float) + 64;
}

extern "C" int polus_crf_nll(int dtype, const float* potentials, const int32_t* tags, const int32_t* lengths,
                             const float* trans, const float* sample_w, float* loss, void* dpot,
                             float* dtrans, int accumulate, int B, int S, int C,
                             void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(potentials && tags && trans && loss && dpot && dtrans, "polus_crf_nll: null pointer");
    POLUS_REQUIRE(B > 0 && S > 0 && C > 0 && C <= CRF_MAXC, "polus_crf_nll: need 0 < C <= %d (got %d)", CRF_MAXC, C);
    if (!workspace || workspace_bytes < polus_crf_workspace_bytes(B, S, C)) { polus_set_error("polus_crf_nll: workspace too small"); return POLUS_ERR_WORKSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* alpha = static_cast<float*>(workspace);
    float* nll_b = alpha + (size_t)B * S * C;
    float* dtb = nll_b + B;
    int blocks = (B + 63) / 64;
    if (dtype == POLUS_BF16)
        hipLaunchKernelGGL(crf_nll_kernel<bf16_t>, dim3(blocks), dim3(64), 0, st, potentials, tags, lengths, trans, sample_w, nll_b, (bf16_t*)dpot, dtb, alpha, B, S, C);
    else if (dtype == POLUS_F32)
        hipLaunchKernelGGL(crf_nll_kernel<float>, dim3(blocks), dim3(64), 0, st, potentials, tags, lengths, trans, sample_w, nll_b, (float*)dpot, dtb, alpha, B, S, C);
    else POLUS_FAIL("polus_crf_nll: bad dtype");
    POLUS_CHECK_LAUNCH("polus_crf_nll");
    hipLaunchKernelGGL(crf_finalize_kernel, dim3(1), dim3(256), 0, st, nll_b, dtb, B, C, loss, dtrans, accumulate);
    POLUS_CHECK_LAUNCH("polus_crf_nll(finalize)");
    return POLUS_OK;
}

extern "C" int polus_crf_viterbi(const float* potentials, const int32_t* lengths, const float* trans,
                                 int32_t* out_tags, int B, int S, int C,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    POLUS_REQUIRE(potentials && trans && out_tags, "polus_crf_viterbi: null pointer");
    POLUS_REQUIRE(B > 0 && S > 0 && C > 0 && C <= CRF_MAXC, "polus_crf_viterbi: need 0 < C <= %d", CRF_MAXC);
    if (!workspace || workspace_bytes < polus_crf_workspace_bytes(B, S, C)) { polus_set_error("polus_crf_viterbi: workspace too small"); return POLUS_ERR_WORKSPACE; }
    hipLaunchKernelGGL(crf_viterbi_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                       potentials, lengths, trans, out_tags, static_cast<int32_t*>(workspace), B, S, C);
    POLUS_CHECK_LAUNCH("polus_crf_viterbi");
    return POLUS_OK;
}
