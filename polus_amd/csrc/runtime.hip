// Error plumbing, device query and the small elementwise utilities of the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void polus_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* polus_last_error(void) { return g_err; }

static PolusCfg g_cfg;
static bool g_cfg_ready = false;
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
static void read_cfg() {
    g_cfg.gemm_pp = env_int("POLUS_GEMM_PP", 0);
    g_cfg.gemm_v1 = getenv("POLUS_GEMM_V1") != nullptr;
    g_cfg.dw_ungrouped = getenv("POLUS_DW_UNGROUPED") != nullptr;
    g_cfg.gemm_order = env_int("POLUS_GEMM_ORDER", 4);
    g_cfg.reserve_cus = env_int("POLUS_GEMM_RESERVE_CUS", 0);
    g_cfg.gemm_persist = env_int("POLUS_GEMM_PERSIST", 1);
    g_cfg.dw_fused_reduce = env_int("POLUS_DW_FUSED_REDUCE", 1);
    g_cfg.attn_fused = env_int("POLUS_ATTN_FUSED", 1);
    g_cfg.attn_bwd_kres = env_int("POLUS_ATTN_BWD_KRES", 1);
    g_cfg.attn_debug = env_int("POLUS_ATTN_DEBUG", 0);
    g_cfg.ln_halfwave = env_int("POLUS_LN_HALFWAVE", 1);
    g_cfg.gemm_auto_split = env_int("POLUS_GEMM_AUTO_SPLIT", 1);
    g_cfg.gemm_ring128 = env_int("POLUS_GEMM_RING128", 0);
    g_cfg.ln_bwd_blocks = env_int("POLUS_LN_BWD_BLOCKS", 512);
    g_cfg.ln_fin_single = env_int("POLUS_LN_FIN_SINGLE", 512);
    g_cfg.dw_streamk = env_int("POLUS_DW_STREAMK", 0);
    g_cfg.dw_sk_delta = env_int("POLUS_DW_SK_DELTA", 2);
    g_cfg.dw_sk_cus = env_int("POLUS_DW_SK_CUS", 0);
    g_cfg_ready = true;
}
const PolusCfg& polus_cfg() {
    if (!g_cfg_ready) read_cfg();
    return g_cfg;
}
extern "C" int polus_reload_env(void) { read_cfg(); return POLUS_OK; }

// The CU reserve only pays while a collective's channel kernels are resident: the trainer switches it on around backward
// (where the bucketed exchange runs) and off for the forward pass, which shares the chip with nothing.
int polus_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
static int g_reserve_on = 1;
extern "C" int polus_set_reserve_active(int on) { g_reserve_on = on != 0; return POLUS_OK; }
int polus_reserved_cus() { return g_reserve_on ? polus_cfg().reserve_cus : 0; }

static const PolusDyn* g_dyn = nullptr;
const PolusDyn* polus_dyn() { return g_dyn; }
extern "C" int polus_set_dynamic_params(const void* dev_block16) {
    POLUS_REQUIRE(dev_block16 == nullptr || polus_aligned16(dev_block16), "polus_set_dynamic_params: the block must be 16-byte aligned");
    g_dyn = static_cast<const PolusDyn*>(dev_block16);
    return POLUS_OK;
}
extern "C" int polus_abi_version(void) { return POLUS_ABI_VERSION; }

extern "C" int polus_device_info(int* n_cu, int* lds_bytes_per_cu, char* arch, int arch_len) {
    int dev = 0;
    POLUS_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    POLUS_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return POLUS_OK;
}

namespace {

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i4 < n; i4 += stride) {
        if (i4 + 4 <= n) {
            float v[4];
            load4<S>(src + i4, v);
            store4<D>(dst + i4, v);
        } else {
            for (int64_t j = i4; j < n; ++j) dst[j] = from_f<D>(to_f<S>(src[j]));
        }
    }
}

__global__ void scale_kernel(float* __restrict__ x, float a, int64_t n) {
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i4 < n; i4 += stride) {
        if (i4 + 4 <= n) {
            float4 v = *reinterpret_cast<float4*>(x + i4);
            v.x *= a; v.y *= a; v.z *= a; v.w *= a;
            *reinterpret_cast<float4*>(x + i4) = v;
        } else {
            for (int64_t j = i4; j < n; ++j) x[j] *= a;
        }
    }
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u, T* __restrict__ du, int64_t n, int act) {
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i4 < n; i4 += stride) {
        if (i4 + 4 <= n) {
            float a[4], b[4];
            load4<T>(dy + i4, a);
            load4<T>(u + i4, b);
            apply_act_grad_n<4>(act, a, b);
            store4<T>(du + i4, a);
        } else {
            for (int64_t j = i4; j < n; ++j) { float a1[1] = {to_f<T>(dy[j])}, b1[1] = {to_f<T>(u[j])}; apply_act_grad_n<1>(act, a1, b1); du[j] = from_f<T>(a1[0]); }
        }
    }
}

// dst[c][r] = src[r][c] for 2-byte elements, 64x64 tiles through LDS (bf16 weight shadows)
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ src,
                                                         unsigned short* __restrict__ dst, int R, int C) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? src[(long)r * C + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        int c = c0 + k, r = r0 + tx;
        if (c < C && r < R) dst[(long)c * R + r] = tile[tx][k];
    }
}

// Same for many matrices that live at the same element offsets of two flat arenas (the bf16 weight
// shadow and its transposed twin): one launch per optimizer step instead of one per weight.
// segs[s] = {element offset, rows, cols, first tile}; tiles are 64x64, row-major over the matrix.
__global__ __launch_bounds__(256) void transpose16_batched_kernel(const unsigned short* __restrict__ src,
                                                                 unsigned short* __restrict__ dst,
                                                                 const long* __restrict__ segs, int nseg) {
    __shared__ unsigned short tile[64][66];
    const int b = blockIdx.x;
    int s = 0;
    while (s + 1 < nseg && segs[4 * (s + 1) + 3] <= b) ++s;     // uniform scan (tens of entries)
    const long off = segs[4 * s];
    const int R = (int)segs[4 * s + 1], C = (int)segs[4 * s + 2];
    const int t = b - (int)segs[4 * s + 3], tc = (C + 63) / 64;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const unsigned short* sp = src + off;
    unsigned short* dp = dst + off;
    if (((R | C) & 3) == 0) {
        // 8-byte accesses: thread t reads 4 consecutive columns of row t/16 (+16 per pass), writes 4
        // consecutive rows of output row t/16 (+16 per pass); arena offsets are 64-element aligned
        const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + rr + 16 * k, c = c0 + 4 * q;
            ushort4 v = make_ushort4(0, 0, 0, 0);
            if (r < R && c < C) v = *reinterpret_cast<const ushort4*>(sp + (long)r * C + c);
            tile[rr + 16 * k][4 * q] = v.x; tile[rr + 16 * k][4 * q + 1] = v.y;
            tile[rr + 16 * k][4 * q + 2] = v.z; tile[rr + 16 * k][4 * q + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + rr + 16 * k, r = r0 + 4 * q;
            if (c < C && r < R) {
                ushort4 v = make_ushort4(tile[4 * q][rr + 16 * k], tile[4 * q + 1][rr + 16 * k],
                                         tile[4 * q + 2][rr + 16 * k], tile[4 * q + 3][rr + 16 * k]);
                *reinterpret_cast<ushort4*>(dp + (long)c * R + r) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? sp[(long)r * C + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        int c = c0 + k, r = r0 + tx;
        if (c < C && r < R) dp[(long)c * R + r] = tile[tx][k];
    }
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, unsigned seed, unsigned thresh, float inv,
                               const PolusDyn* dyn) {
    seed = polus_eff_seed(seed, dyn);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = polus_keep(seed, (unsigned)i, thresh) ? from_f<T>(to_f<T>(x[i]) * inv) : from_f<T>(0.f);
}

__global__ void dropout_mask_kernel(unsigned seed, unsigned thresh, unsigned idx0, int64_t n, uint8_t* __restrict__ mask,
                                    const PolusDyn* dyn) {
    seed = polus_eff_seed(seed, dyn);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) mask[i] = polus_keep(seed, idx0 + (unsigned)i, thresh) ? 1 : 0;
}

inline int stream_grid(int64_t n) {
    int64_t b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;  // memory-bound: cap and grid-stride
    return (int)b;
}

}  // namespace

extern "C" int polus_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream) {
    POLUS_REQUIRE(src && dst && n >= 0, "polus_cast: bad arguments");
    if (n == 0) return POLUS_OK;
    POLUS_REQUIRE(((uintptr_t)src % (4 * polus_dtype_size(src_dtype))) == 0 &&
                  ((uintptr_t)dst % (4 * polus_dtype_size(dst_dtype))) == 0,
                  "polus_cast: pointers must be 4-element aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(stream_grid(n)), block(256);
    if (src_dtype == POLUS_F32 && dst_dtype == POLUS_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == POLUS_BF16 && dst_dtype == POLUS_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == POLUS_F32 && dst_dtype == POLUS_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, st, (const float*)src, (float*)dst, n);
    else if (src_dtype == POLUS_BF16 && dst_dtype == POLUS_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    else POLUS_FAIL("polus_cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
    POLUS_CHECK_LAUNCH("polus_cast");
    return POLUS_OK;
}

extern "C" int polus_scale(float* x, float a, int64_t n, void* stream) {
    POLUS_REQUIRE(x && n >= 0, "polus_scale: bad arguments");
    if (n == 0) return POLUS_OK;
    POLUS_REQUIRE(((uintptr_t)x % 16) == 0, "polus_scale: pointer must be 16-byte aligned");
    hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, a, n);
    POLUS_CHECK_LAUNCH("polus_scale");
    return POLUS_OK;
}

extern "C" int polus_act_bwd(int dtype, const void* dy, const void* u, void* du, int64_t n, int act, void* stream) {
    POLUS_REQUIRE(dy && u && du && n >= 0, "polus_act_bwd: bad arguments");
    if (n == 0) return POLUS_OK;
    size_t es = polus_dtype_size(dtype);
    POLUS_REQUIRE(((uintptr_t)dy % (4 * es)) == 0 && ((uintptr_t)u % (4 * es)) == 0 && ((uintptr_t)du % (4 * es)) == 0,
                  "polus_act_bwd: pointers must be 4-element aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == POLUS_BF16)
        hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(stream_grid(n)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)du, n, act);
    else if (dtype == POLUS_F32)
        hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(stream_grid(n)), dim3(256), 0, st, (const float*)dy, (const float*)u, (float*)du, n, act);
    else POLUS_FAIL("polus_act_bwd: bad dtype");
    POLUS_CHECK_LAUNCH("polus_act_bwd");
    return POLUS_OK;
}

extern "C" int polus_transpose_bf16_batched(const void* src_base, void* dst_base, const void* segs_dev, int nseg,
                                            int total_tiles, void* stream) {
    POLUS_REQUIRE(src_base && dst_base && segs_dev && nseg > 0 && total_tiles > 0, "polus_transpose_bf16_batched: bad arguments");
    hipLaunchKernelGGL(transpose16_batched_kernel, dim3(total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const unsigned short*)src_base, (unsigned short*)dst_base, (const long*)segs_dev, nseg);
    POLUS_CHECK_LAUNCH("polus_transpose_bf16_batched");
    return POLUS_OK;
}

extern "C" int polus_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream) {
    POLUS_REQUIRE(src && dst && rows > 0 && cols > 0, "polus_transpose_bf16: bad arguments");
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const unsigned short*)src, (unsigned short*)dst, rows, cols);
    POLUS_CHECK_LAUNCH("polus_transpose_bf16");
    return POLUS_OK;
}

extern "C" int polus_dropout_mask(uint32_t seed, float drop_p, uint32_t idx0, int64_t n, uint8_t* mask, void* stream) {
    POLUS_REQUIRE(mask && n >= 0 && drop_p >= 0.0f && drop_p < 1.0f, "polus_dropout_mask: bad arguments");
    if (n == 0) return POLUS_OK;
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((int)b), dim3(256), 0, static_cast<hipStream_t>(stream),
                       seed, polus_drop_thresh(drop_p), idx0, n, mask, polus_dyn());
    POLUS_CHECK_LAUNCH("polus_dropout_mask");
    return POLUS_OK;
}

extern "C" int polus_dropout(int dtype, const void* x, void* y, int64_t n, float drop_p, uint32_t seed, void* stream) {
    POLUS_REQUIRE(x && y && n >= 0 && n < (1LL << 32) && drop_p >= 0.0f && drop_p < 1.0f, "polus_dropout: bad arguments");
    if (n == 0) return POLUS_OK;
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned th = drop_p > 0.f ? polus_drop_thresh(drop_p) : 0u;
    const float inv = 1.0f / (1.0f - drop_p);
    if (dtype == POLUS_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3((int)b), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n, seed, th, inv, polus_dyn());
    else if (dtype == POLUS_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3((int)b), dim3(256), 0, st, (const float*)x, (float*)y, n, seed, th, inv, polus_dyn());
    else POLUS_FAIL("polus_dropout: bad dtype");
    POLUS_CHECK_LAUNCH("polus_dropout");
    return POLUS_OK;
}
