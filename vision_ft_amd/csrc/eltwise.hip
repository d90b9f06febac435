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
            da[i] = d[i] * gelu_f(g[i]);
            dg[i] = d[i] * a[i] * dgelu_f(g[i]);
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
