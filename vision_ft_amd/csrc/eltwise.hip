// Fused element-wise kernels of the transformer blocks, bf16 in/out, fp32 math — gfx950.
// All are single-pass HBM streams with 16-byte accesses.
//
//   GEGLU            hs * gelu(gate)             src/models/sdxl/denoiser.py:183-186 (GeGLU.forward)
//   SwiGLU           silu(a) * b                 src/models/auraflow/denoiser.py:113-119, lumina2/denoiser.py:212-219
//   gated residual   res + g[b,:] * y            auraflow/denoiser.py:408-422 ; with tanh(g): lumina2/denoiser.py:303,311
#include "vft_common.h"

namespace {

__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}

// proj [M][2F] -> out [M][F]
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* __restrict__ proj, bf16_t* __restrict__ out,
                                                        int64_t M, int F) {
    const int64_t nvec = M * (F / 8);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        const int64_t m = e / (F / 8);
        const int c = (int)(e % (F / 8)) * 8;
        float a[8], g[8], o[8];
        ld8(proj + m * 2 * F + c, a);
        ld8(proj + m * 2 * F + F + c, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = a[i] * gelu_f(g[i]);
        st8(out + m * F + c, o);
    }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ proj, const bf16_t* __restrict__ dy,
                                                        bf16_t* __restrict__ dproj, int64_t M, int F) {
    const int64_t nvec = M * (F / 8);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        const int64_t m = e / (F / 8);
        const int c = (int)(e % (F / 8)) * 8;
        float a[8], g[8], d[8], da[8], dg[8];
        ld8(proj + m * 2 * F + c, a);
        ld8(proj + m * 2 * F + F + c, g);
        ld8(dy + m * F + c, d);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float gl, dgl;
            gelu_both(g[i], gl, dgl);
            da[i] = d[i] * gl;
            dg[i] = d[i] * a[i] * dgl;
        }
        st8(dproj + m * 2 * F + c, da);
        st8(dproj + m * 2 * F + F + c, dg);
    }
}


__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                         bf16_t* __restrict__ out, int64_t nvec) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        float x[8], y[8], o[8];
        ld8(a + e * 8, x);
        ld8(b + e * 8, y);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = silu_f(x[i]) * y[i];
        st8(out + e * 8, o);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                         const bf16_t* __restrict__ dy, bf16_t* __restrict__ da,
                                                         bf16_t* __restrict__ db, int64_t nvec) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        float x[8], y[8], d[8], ox[8], oy[8];
        ld8(a + e * 8, x);
        ld8(b + e * 8, y);
        ld8(dy + e * 8, d);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ox[i] = d[i] * y[i] * dsilu_f(x[i]);
            oy[i] = d[i] * silu_f(x[i]);
        }
        st8(da + e * 8, ox);
        st8(db + e * 8, oy);
    }
}

// out[b,s,:] = (res ? res[b,s,:] : 0) + f(gate[b,:]) * y[b,s,:],  f = tanh or identity
__global__ __launch_bounds__(256) void gate_residual_kernel(const bf16_t* __restrict__ res, const bf16_t* __restrict__ gate,
                                                            const bf16_t* __restrict__ y, bf16_t* __restrict__ out,
                                                            int64_t rows, int C, int rows_per_batch, int use_tanh) {
    const int64_t nvec = rows * (C / 8);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / (C / 8);
        const int c = (int)(e % (C / 8)) * 8;
        const int64_t b = row / rows_per_batch;
        float g[8], v[8], o[8];
        ld8(gate + b * C + c, g);
        ld8(y + row * C + c, v);
        if (res) ld8(res + row * C + c, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gg = use_tanh ? tanhf(g[i]) : g[i];
            o[i] = (res ? o[i] : 0.f) + gg * v[i];
        }
        st8(out + row * C + c, o);
    }
}

unsigned grid_for(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int vft_geglu_fwd(const void* proj, void* out, int64_t M, int F, void* stream) {
    VFT_CHECK_ARG(proj && out && M > 0 && F > 0 && F % 8 == 0, "vft_geglu_fwd: bad arguments (F must be a multiple of 8)");
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(M * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)proj, (bf16_t*)out, M, F);
    VFT_CHECK_LAUNCH("vft_geglu_fwd");
    return 0;
}

extern "C" int vft_geglu_bwd(const void* proj, const void* dy, void* dproj, int64_t M, int F, void* stream) {
    VFT_CHECK_ARG(proj && dy && dproj && M > 0 && F > 0 && F % 8 == 0, "vft_geglu_bwd: bad arguments");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(M * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)proj, (const bf16_t*)dy, (bf16_t*)dproj, M, F);
    VFT_CHECK_LAUNCH("vft_geglu_bwd");
    return 0;
}

extern "C" int vft_swiglu_fwd(const void* a, const void* b, void* out, int64_t n, void* stream) {
    VFT_CHECK_ARG(a && b && out && n > 0 && n % 8 == 0, "vft_swiglu_fwd: bad arguments (n must be a multiple of 8)");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)out, n / 8);
    VFT_CHECK_LAUNCH("vft_swiglu_fwd");
    return 0;
}

extern "C" int vft_swiglu_bwd(const void* a, const void* b, const void* dy, void* da, void* db, int64_t n,
                              void* stream) {
    VFT_CHECK_ARG(a && b && dy && da && db && n > 0 && n % 8 == 0, "vft_swiglu_bwd: bad arguments");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, (const bf16_t*)dy, (bf16_t*)da, (bf16_t*)db, n / 8);
    VFT_CHECK_LAUNCH("vft_swiglu_bwd");
    return 0;
}

extern "C" int vft_gate_residual(const void* residual, const void* gate, const void* y, void* out, int64_t rows, int C,
                                 int rows_per_batch, int use_tanh, void* stream) {
    VFT_CHECK_ARG(gate && y && out && rows > 0 && C > 0 && C % 8 == 0 && rows_per_batch > 0, "vft_gate_residual: bad arguments");
    hipLaunchKernelGGL(gate_residual_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)residual, (const bf16_t*)gate, (const bf16_t*)y, (bf16_t*)out, rows, C,
                       rows_per_batch, use_tanh);
    VFT_CHECK_LAUNCH("vft_gate_residual");
    return 0;
}


// ---------------------------------------------------------------------------------------------
// dst[C][R] = src[R][C]^T (bf16): the per-step transposed copy of a TRAINABLE projection weight, on which the data gradient
// dx = dy . W runs as the same NT GEMM (kernels._transposed_weight; the frozen weights' copies are made once).  64 x 64 tiles
// through LDS (rows padded by one dword: the column reads hit 32 different banks), 16-byte global accesses on both sides.
// Replaces `weight.t().contiguous()`: ATen's generic strided copy ran the 323 weights of the SDXL full fine-tune at
// 0.3 TB/s — 13 ms per step.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int C,
                                                        int64_t lds_, int64_t ldd) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + 256 * q, row = idx >> 3, ch = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + row < R && c0 + ch * 8 < C) v = *reinterpret_cast<const u32x4*>(src + (int64_t)(r0 + row) * lds_ + c0 + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[row][ch * 8 + 2 * e] = (unsigned short)(v[e] & 0xffffu);
            tile[row][ch * 8 + 2 * e + 1] = (unsigned short)(v[e] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + 256 * q, col = idx >> 3, ch = idx & 7;  // output row = source column
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (unsigned)tile[ch * 8 + 2 * e][col] | ((unsigned)tile[ch * 8 + 2 * e + 1][col] << 16);
        if (c0 + col < C && r0 + ch * 8 < R) *reinterpret_cast<u32x4*>(dst + (int64_t)(c0 + col) * ldd + r0 + ch * 8) = v;
    }
}
}  // namespace

extern "C" int vft_transpose_bf16(const void* src, void* dst, int R, int C, int64_t ld_src, int64_t ld_dst, void* stream) {
    VFT_CHECK_ARG(src && dst, "vft_transpose_bf16: null pointer");
    VFT_CHECK_ARG(R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ld_src >= C && ld_dst >= R,
                  "vft_transpose_bf16: R, C and both row strides must be positive multiples of 8 (R=%d C=%d)", R, C);
    VFT_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "vft_transpose_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, R, C, ld_src, ld_dst);
    VFT_CHECK_LAUNCH("vft_transpose_bf16");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// mean(x^2) of a dense fp32 buffer — the reduction of the reference's MSE losses (src/modules/loss/diffusion.py:62-68,
// loss/flow_match.py:85-96: F.mse_loss(..., reduction="mean") on fp32 tensors).  Two launches, no atomics, no memset:
// workgroup g sums a fixed contiguous chunk in a fixed order into partial[g]; one workgroup folds the partials in index
// order and scales by 1 / n.  Bit-reproducible, and — the reason it exists (round 4) — safe inside a replayed HIP graph:
// ATen's full reduction is ONE launch whose workgroups meet through a semaphore buffer zeroed by a memset node in front
// of it; in the captured SDXL step that reduction returned garbage (negative "MSE" values) for every replay after the
// process had synchronised an idle stream, while every other tensor of the step stayed bit-identical to the eager run.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int MSQ_BLOCKS = 1024;
// T = float: 4 elements per 16-byte load; T = bf16_t: 8
template <class T>
__global__ __launch_bounds__(256) void mean_sq_partial_kernel(const T* __restrict__ x, float* __restrict__ partial, int64_t n) {
    constexpr int V = 16 / sizeof(T);
    __shared__ float red[256];
    const int64_t chunk = ((n + MSQ_BLOCKS - 1) / MSQ_BLOCKS + V - 1) / V * V;  // multiple of V elements: 16-byte loads stay aligned
    const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
    float s = 0.f;
    for (int64_t i = b0 + V * (int64_t)threadIdx.x; i < b1; i += 256 * V) {
        if (i + V <= b1) {
            if constexpr (sizeof(T) == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) s += v[j] * v[j];
            } else {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)v[j] * (float)v[j];
            }
        } else {
            for (int64_t j = i; j < b1; ++j) s += (float)x[j] * (float)x[j];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] (+)= scale * sum of the partials (accumulate: several buffers folded into one sum, launch after launch)
__global__ __launch_bounds__(256) void mean_sq_final_kernel(const float* __restrict__ partial, float* __restrict__ out, float scale, int accumulate) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < MSQ_BLOCKS; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale + (accumulate ? out[0] : 0.f);
}
}  // namespace

extern "C" int64_t vft_mean_sq_scratch_floats(void) { return MSQ_BLOCKS; }

extern "C" int vft_mean_sq(const float* x, float* out, float* scratch, int64_t n, void* stream) {
    VFT_CHECK_ARG(x && out && scratch, "vft_mean_sq: null pointer");
    VFT_CHECK_ARG(n > 0 && ((uintptr_t)x & 15) == 0, "vft_mean_sq: needs n > 0 and a 16-byte aligned input (n=%lld)", (long long)n);
    hipLaunchKernelGGL(mean_sq_partial_kernel<float>, dim3(MSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, x, scratch, n);
    hipLaunchKernelGGL(mean_sq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, out, (float)(1.0 / (double)n), 0);
    VFT_CHECK_LAUNCH("vft_mean_sq");
    return 0;
}

// out[0] (+)= sum(x[i]^2) over a dense bf16 buffer: the squared L2 norm of a flat gradient buffer, i.e. the reduction of
// gradient-norm clipping (reference: ModelForTraining.after_backward -> clip_grad_norm_, src/models/for_training.py:171-189),
// on the same two-launch scheme — for the same reason and because one pass over ONE buffer replaces a
// multi-tensor pass over 1 400 gradient views.
extern "C" int vft_sum_sq_bf16(const void* x, float* out, float* scratch, int64_t n, int accumulate, void* stream) {
    VFT_CHECK_ARG(x && out && scratch, "vft_sum_sq_bf16: null pointer");
    VFT_CHECK_ARG(n > 0 && ((uintptr_t)x & 15) == 0, "vft_sum_sq_bf16: needs n > 0 and a 16-byte aligned input (n=%lld)", (long long)n);
    hipLaunchKernelGGL(mean_sq_partial_kernel<bf16_t>, dim3(MSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scratch, n);
    hipLaunchKernelGGL(mean_sq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, out, 1.f, accumulate);
    VFT_CHECK_LAUNCH("vft_sum_sq_bf16");
    return 0;
}
