// GroupNorm (+ fused SiLU) over NCHW or NHWC bf16 activations, fp32 statistics — gfx950.
//
// Replaces cuDNN/ATen GroupNorm followed by a separate SiLU in the SDXL ResidualBlock
// (src/models/sdxl/denoiser.py:511-551, GN32 eps 1e-5 -> SiLU -> conv) and the SpatialTransformer
// input norm (:303, eps 1e-6, no activation).
//
// Two HBM passes per direction (statistics, then apply): a group of an NCHW tensor is one contiguous
// run of (C/G)·H·W elements, so both passes stream with 16-byte accesses.  The statistics pass runs
// one workgroup per (n, c) plane (NCHW) or per row chunk (NHWC) and writes its partial sums to a scratch
// buffer; a tiny finalize kernel adds them in a FIXED order, so statistics, dx and the affine gradients are
// bit-reproducible from run to run (no floating-point atomics anywhere) and a checkpoint recompute sees the
// statistics the forward saw.  Algorithmic bytes: forward 3·N·2 (two reads + one write of the tensor),
// backward 5·N·2.
#include "vft_common.h"

namespace {

// V-wide (V = 8, 4, 2, 1 elements) loads/stores: V is the largest power of two dividing H*W, so a
// vector never straddles two (n, c) planes
template <int V>
__device__ __forceinline__ void ldv(const bf16_t* p, float (&v)[V]) {
    typedef __attribute__((ext_vector_type(V))) __bf16 vec_t;
    if constexpr (V == 1) {
        v[0] = (float)p[0];
    } else {
        vec_t t = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = (float)t[i];
    }
}
template <int V>
__device__ __forceinline__ void stv(bf16_t* p, const float (&v)[V]) {
    typedef __attribute__((ext_vector_type(V))) __bf16 vec_t;
    if constexpr (V == 1) {
        p[0] = (bf16_t)v[0];
    } else {
        vec_t t;
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = (bf16_t)v[i];
        *reinterpret_cast<vec_t*>(p) = t;
    }
}

// part[(n*C+c)*2 + {0,1}] = (Σx, Σx²) of plane (n, c); gn_finalize_kernel folds the planes of a group in channel order
template <int V>
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int C,
                                                       int HW, int cpg) {
    __shared__ float red[4];
    const int plane = blockIdx.x;  // n*C + c
    const int n = plane / C, c = plane % C;
    const bf16_t* xp = x + (int64_t)plane * HW;
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x * V; e < HW; e += 256 * V) {
        float v[V];
        ldv<V>(xp + e, v);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            s1 += v[i];
            s2 += v[i] * v[i];
        }
    }
    s1 = block_sum<4>(s1, red);
    s2 = block_sum<4>(s2, red);
    if (threadIdx.x == 0) {
        part[(int64_t)plane * 2] = s1;
        part[(int64_t)plane * 2 + 1] = s2;
    }
    (void)n, (void)c, (void)cpg;
}

// out[row*2 + k] = Σ_{j < nparts} part[(row*nparts + j)*2 + k]   (row = (n, g); parts = the group's planes (NCHW) or
// the image's row chunks laid out [n][g][chunk] (NHWC)), added in index order: deterministic
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int64_t rows2, int nparts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows2) return;
    const int64_t row = i >> 1;
    const int k = (int)(i & 1);
    float v = 0.f;
    for (int j = 0; j < nparts; ++j) v += part[(row * nparts + j) * 2 + k];
    out[i] = v;
}

// affine gradients: dw[c] = Σ_blk cpart[blk][0][c], db[c] = Σ_blk cpart[blk][1][c].  A workgroup owns 32 channels; its 8
// thread rows each add a contiguous slice of the blocks in index order, then the 8 slice sums are added in slice
// order: a fixed association for a given (nblk, C), hence deterministic, with 8x the parallelism of one thread per channel
__global__ __launch_bounds__(256) void gn_finalize_affine_kernel(const float* __restrict__ cpart, float* __restrict__ dw,
                                                                 float* __restrict__ db, int nblk, int C) {
    __shared__ float red[2][8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int per = (nblk + 7) / 8;
    const int j0 = sl * per, j1 = min(nblk, j0 + per);
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int j = j0; j < j1; ++j) {
            a += cpart[((int64_t)j * 2) * C + c];
            b += cpart[((int64_t)j * 2 + 1) * C + c];
        }
    red[0][sl][cl] = a;
    red[1][sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sa += red[0][q][cl];
            sb += red[1][q][cl];
        }
        if (dw) dw[c] = sa;
        if (db) db[c] = sb;
    }
}

template <int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                       const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ y, int C, int HW, int cpg, float eps,
                                                       int silu, int64_t nvec) {
    const int G = C / cpg;
    const float invL = 1.f / ((float)cpg * (float)HW);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        const int64_t idx = e * V;
        const int64_t plane = idx / HW;
        const int c = (int)(plane % C);
        const int64_t n = plane / C;
        const float* st = stats + (n * G + c / cpg) * 2;
        const float mean = st[0] * invL;
        const float rstd = rsqrtf(fmaxf(st[1] * invL - mean * mean, 0.f) + eps);
        const float wc = w ? (float)w[c] : 1.f, bc = b ? (float)b[c] : 0.f;
        const float a = rstd * wc, o = bc - mean * a;
        float v[V];
        ldv<V>(x + idx, v);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float z = v[i] * a + o;
            v[i] = silu ? silu_f(z) : z;
        }
        stv<V>(y + idx, v);
    }
}

// part[(n*C+c)*2 + {0,1}] = (Σ g, Σ g·xhat) of plane (n, c), g = dz·w_c, dz = dy·silu'(z) (or dy);
// optionally cpart[n][0][c] = Σ dz·xhat, cpart[n][1][c] = Σ dz (the affine gradients' per-image partials)
template <int V>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           const float* __restrict__ stats, const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ b, float* __restrict__ part,
                                                           float* __restrict__ cpart, int C, int HW,
                                                           int cpg, float eps, int silu) {
    __shared__ float red[4];
    const int plane = blockIdx.x;
    const int n = plane / C, c = plane % C;
    const int G = C / cpg;
    const float invL = 1.f / ((float)cpg * (float)HW);
    const float* st = stats + ((int64_t)n * G + c / cpg) * 2;
    const float mean = st[0] * invL;
    const float rstd = rsqrtf(fmaxf(st[1] * invL - mean * mean, 0.f) + eps);
    const float wc = w ? (float)w[c] : 1.f, bc = b ? (float)b[c] : 0.f;
    const bf16_t* xp = x + (int64_t)plane * HW;
    const bf16_t* dp = dy + (int64_t)plane * HW;
    float sz = 0.f, szx = 0.f;
    for (int e = threadIdx.x * V; e < HW; e += 256 * V) {
        float v[V], d[V];
        ldv<V>(xp + e, v);
        ldv<V>(dp + e, d);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float xh = (v[i] - mean) * rstd;
            float dz = d[i];
            if (silu) dz *= dsilu_f(xh * wc + bc);
            sz += dz;
            szx += dz * xh;
        }
    }
    sz = block_sum<4>(sz, red);
    szx = block_sum<4>(szx, red);
    if (threadIdx.x == 0) {
        part[(int64_t)plane * 2] = sz * wc;
        part[(int64_t)plane * 2 + 1] = szx * wc;
        if (cpart) {
            cpart[((int64_t)n * 2) * C + c] = szx;
            cpart[((int64_t)n * 2 + 1) * C + c] = sz;
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                        const float* __restrict__ stats, const float* __restrict__ bstats,
                                                        const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                        bf16_t* __restrict__ dx, int C, int HW, int cpg, float eps,
                                                        int silu, int64_t nvec) {
    const int G = C / cpg;
    const float invL = 1.f / ((float)cpg * (float)HW);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        const int64_t idx = e * V;
        const int64_t plane = idx / HW;
        const int c = (int)(plane % C);
        const int64_t n = plane / C;
        const int64_t row = n * G + c / cpg;
        const float mean = stats[row * 2] * invL;
        const float rstd = rsqrtf(fmaxf(stats[row * 2 + 1] * invL - mean * mean, 0.f) + eps);
        const float mg = bstats[row * 2] * invL, mgx = bstats[row * 2 + 1] * invL;
        const float wc = w ? (float)w[c] : 1.f, bc = b ? (float)b[c] : 0.f;
        float v[V], d[V];
        ldv<V>(x + idx, v);
        ldv<V>(dy + idx, d);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float xh = (v[i] - mean) * rstd;
            float dz = d[i];
            if (silu) dz *= dsilu_f(xh * wc + bc);
            v[i] = rstd * (dz * wc - mg - xh * mgx);
        }
        stv<V>(dx + idx, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Channels-last variant: x[N][HW][C].  The whole UNet runs NHWC on MI355X (MIOpen's implicit-GEMM
// convolutions are NHWC-native, and (N, HW, C) *is* the token layout of the SpatialTransformer, so
// no permute copies are left).  A group is no longer contiguous, so threads own a fixed 8-channel
// vector (cv) and walk rows: block = CV x R threads (CV = C/8), thread (cv, lr) visits rows
// lr, lr+R, ... of the block's row chunk.  Per-channel constants live in registers for the whole
// walk; the per-group sums are folded through LDS and then 2·G global atomics per block.
// Same statistics layout as the NCHW path: stats[(n*G+g)*2 + {0,1}] = (Σx, Σx²).
// ---------------------------------------------------------------------------------------------
struct NhwcGeom {
    int C, HW, cpg, G, CV, R, rows_per_block;
};

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

// Block-level fold of the threads' 8-channel partial pairs, in a fixed order (no atomics): the R row slots of a channel
// are added in slot order, then a group's channels in channel order.  L: (2R + 2)·C floats of LDS.
//   part[((n*G + g) * nchunks + chunk) * 2 + {0,1}] = the block's sums of group g (scale: per-channel factor or null)
//   cpart[(blk*2 + {0,1}) * C + c]                  = the block's per-channel sums (b first, then a), when wanted
__device__ __forceinline__ void fold_block(float* L, const NhwcGeom& gm, int lr, int cv, const float (&a)[8],
                                           const float (&b)[8], const float* wscale, float* part, int64_t part_row0,
                                           int nchunks, int chunk, float* cpart, int64_t blk) {
    const int C = gm.C, R = gm.R;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        L[(lr)*C + cv * 8 + i] = a[i];
        L[(R + lr) * C + cv * 8 + i] = b[i];
    }
    __syncthreads();
    float* pc = L + 2 * R * C;  // [2][C]
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sa = 0.f, sb = 0.f;
        for (int q = 0; q < R; ++q) {
            sa += L[q * C + c];
            sb += L[(R + q) * C + c];
        }
        if (cpart) {
            cpart[(blk * 2) * C + c] = sb;
            cpart[(blk * 2 + 1) * C + c] = sa;
        }
        const float wc = wscale ? wscale[c] : 1.f;
        pc[c] = sa * wc;
        pc[C + c] = sb * wc;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < gm.G; g += blockDim.x) {
        float sa = 0.f, sb = 0.f;
        for (int j = 0; j < gm.cpg; ++j) {
            sa += pc[g * gm.cpg + j];
            sb += pc[C + g * gm.cpg + j];
        }
        float* o = part + ((part_row0 + g) * nchunks + chunk) * 2;
        o[0] = sa;
        o[1] = sb;
    }
}

__global__ __launch_bounds__(320) void gn_nhwc_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                            NhwcGeom gm) {
    extern __shared__ float gacc[];  // (2R + 2)·C
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int lr = tid / gm.CV, cv = tid - lr * gm.CV;
    const int r0 = blockIdx.x * gm.rows_per_block;
    const int r1 = min(r0 + gm.rows_per_block, gm.HW);
    const bf16_t* xp = x + ((int64_t)n * gm.HW) * gm.C + cv * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    int r = r0 + lr;
    for (; r + 3 * gm.R < r1; r += 4 * gm.R) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8(xp + (int64_t)(r + u * gm.R) * gm.C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s1[i] += v[u][i];
                s2[i] += v[u][i] * v[u][i];
            }
    }
    for (; r < r1; r += gm.R) {
        float v[8];
        ld8(xp + (int64_t)r * gm.C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s1[i] += v[i];
            s2[i] += v[i] * v[i];
        }
    }
    fold_block(gacc, gm, lr, cv, s1, s2, nullptr, part, (int64_t)n * gm.G, gridDim.x, blockIdx.x, nullptr, 0);
}

// per-channel constants of thread's 8 channels
struct ChanConst {
    float mean[8], rstd[8], w[8], b[8];
};
__device__ __forceinline__ void load_chan_const(ChanConst& k, const float* stats, const bf16_t* w, const bf16_t* b, int n,
                                                int c0, const NhwcGeom& gm, float eps) {
    const float invL = 1.f / ((float)gm.cpg * (float)gm.HW);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c0 + i;
        const float* st = stats + ((int64_t)n * gm.G + c / gm.cpg) * 2;
        k.mean[i] = st[0] * invL;
        k.rstd[i] = rsqrtf(fmaxf(st[1] * invL - k.mean[i] * k.mean[i], 0.f) + eps);
        k.w[i] = w ? (float)w[c] : 1.f;
        k.b[i] = b ? (float)b[c] : 0.f;
    }
}

__global__ __launch_bounds__(320) void gn_nhwc_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                            const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                            bf16_t* __restrict__ y, NhwcGeom gm, float eps, int silu) {
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int lr = tid / gm.CV, cv = tid - lr * gm.CV;
    ChanConst k;
    load_chan_const(k, stats, w, b, n, cv * 8, gm, eps);
    float a[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = k.rstd[i] * k.w[i];
        o[i] = k.b[i] - k.mean[i] * a[i];
    }
    const int r0 = blockIdx.x * gm.rows_per_block;
    const int r1 = min(r0 + gm.rows_per_block, gm.HW);
    const int64_t base = ((int64_t)n * gm.HW) * gm.C + cv * 8;
    int r = r0 + lr;
    for (; r + 3 * gm.R < r1; r += 4 * gm.R) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8(x + base + (int64_t)(r + u * gm.R) * gm.C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float z = v[u][i] * a[i] + o[i];
                v[u][i] = silu ? silu_f(z) : z;
            }
            st8(y + base + (int64_t)(r + u * gm.R) * gm.C, v[u]);
        }
    }
    for (; r < r1; r += gm.R) {
        float v[8];
        ld8(x + base + (int64_t)r * gm.C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float z = v[i] * a[i] + o[i];
            v[i] = silu ? silu_f(z) : z;
        }
        st8(y + base + (int64_t)r * gm.C, v);
    }
}

__global__ __launch_bounds__(320) void gn_nhwc_bwd_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                const float* __restrict__ stats,
                                                                const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                                float* __restrict__ part, float* __restrict__ cpart,
                                                                NhwcGeom gm, float eps, int silu) {
    extern __shared__ float gacc[];  // (2R + 3)·C: fold_block's image + the channels' weights
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int lr = tid / gm.CV, cv = tid - lr * gm.CV;
    ChanConst k;
    load_chan_const(k, stats, w, b, n, cv * 8, gm, eps);
    const int r0 = blockIdx.x * gm.rows_per_block;
    const int r1 = min(r0 + gm.rows_per_block, gm.HW);
    const int64_t base = ((int64_t)n * gm.HW) * gm.C + cv * 8;
    float sz[8], szx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sz[i] = szx[i] = 0.f;
    for (int r = r0 + lr; r < r1; r += 2 * gm.R) {
        float v[2][8], d[2][8];
        const bool two = r + gm.R < r1;
        ld8(x + base + (int64_t)r * gm.C, v[0]);
        ld8(dy + base + (int64_t)r * gm.C, d[0]);
        if (two) {
            ld8(x + base + (int64_t)(r + gm.R) * gm.C, v[1]);
            ld8(dy + base + (int64_t)(r + gm.R) * gm.C, d[1]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (v[u][i] - k.mean[i]) * k.rstd[i];
                float dz = d[u][i];
                if (silu) dz *= dsilu_f(xh * k.w[i] + k.b[i]);
                sz[i] += dz;
                szx[i] += dz * xh;
            }
        }
    }
    // group sums take g = dz·w_c: the weights as an LDS vector for fold_block's per-channel pass
    float* wl = gacc + (2 * gm.R + 2) * gm.C;
    for (int c = tid; c < gm.C; c += blockDim.x) wl[c] = w ? (float)w[c] : 1.f;
    fold_block(gacc, gm, lr, cv, sz, szx, wl, part, (int64_t)n * gm.G, gridDim.x, blockIdx.x, cpart,
               (int64_t)n * gridDim.x + blockIdx.x);
}

__global__ __launch_bounds__(320) void gn_nhwc_bwd_dx_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ bstats, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ b, bf16_t* __restrict__ dx,
                                                             NhwcGeom gm, float eps, int silu) {
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int lr = tid / gm.CV, cv = tid - lr * gm.CV;
    ChanConst k;
    load_chan_const(k, stats, w, b, n, cv * 8, gm, eps);
    float mg[8], mgx[8];
    {
        const float invL = 1.f / ((float)gm.cpg * (float)gm.HW);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* bs = bstats + ((int64_t)n * gm.G + (cv * 8 + i) / gm.cpg) * 2;
            mg[i] = bs[0] * invL;
            mgx[i] = bs[1] * invL;
        }
    }
    const int r0 = blockIdx.x * gm.rows_per_block;
    const int r1 = min(r0 + gm.rows_per_block, gm.HW);
    const int64_t base = ((int64_t)n * gm.HW) * gm.C + cv * 8;
    for (int r = r0 + lr; r < r1; r += 2 * gm.R) {
        float v[2][8], d[2][8];
        const bool two = r + gm.R < r1;
        ld8(x + base + (int64_t)r * gm.C, v[0]);
        ld8(dy + base + (int64_t)r * gm.C, d[0]);
        if (two) {
            ld8(x + base + (int64_t)(r + gm.R) * gm.C, v[1]);
            ld8(dy + base + (int64_t)(r + gm.R) * gm.C, d[1]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (v[u][i] - k.mean[i]) * k.rstd[i];
                float dz = d[u][i];
                if (silu) dz *= dsilu_f(xh * k.w[i] + k.b[i]);
                v[u][i] = k.rstd[i] * (dz * k.w[i] - mg[i] - xh * mgx[i]);
            }
            st8(dx + base + (int64_t)(r + u * gm.R) * gm.C, v[u]);
        }
    }
}

// block = CV x R threads (<= 320); rows per block sized for >= ~2048 blocks in flight
int nhwc_geom(const char* name, NhwcGeom& gm, int N, int C, int HW, int G, const void* p0, const void* p1, const void* p2) {
    VFT_CHECK_ARG(C % 8 == 0 && C / 8 <= 320, "%s: channels-last path needs C %% 8 == 0 and C <= 2560 (C=%d)", name, C);
    VFT_CHECK_ARG((((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0, "%s: pointers must be 16-byte aligned", name);
    VFT_CHECK_ARG(G * 2 * sizeof(float) <= 48 * 1024, "%s: too many groups (%d)", name, G);
    gm.C = C;
    gm.HW = HW;
    gm.G = G;
    gm.cpg = C / G;
    gm.CV = C / 8;
    gm.R = (256 + gm.CV / 2) / gm.CV;
    while (gm.R > 1 && gm.CV * gm.R > 320) --gm.R;
    if (gm.R < 1) gm.R = 1;
    int64_t per_thread = ((int64_t)N * HW) / ((int64_t)gm.R * 2048);
    if (per_thread < 4) per_thread = 4;
    if (per_thread > 64) per_thread = 64;
    gm.rows_per_block = (int)per_thread * gm.R;
    return 0;
}

unsigned grid_for(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int vec_width(int HW, const void* a, const void* b) {
    int V = 8;
    while (V > 1 && (HW % V != 0 || ((uintptr_t)a & (2 * V - 1)) != 0 || ((uintptr_t)b & (2 * V - 1)) != 0)) V >>= 1;
    return V;
}

int gn_args(const char* name, int N, int C, int HW, int G) {
    VFT_CHECK_ARG(N > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "%s: bad shape N=%d C=%d HW=%d G=%d", name, N, C, HW, G);
    return 0;
}

}  // namespace

// Scratch floats the four entry points need for their partial sums (see the header comment): forward and backward use
// the same amount; `affine` adds the per-block channel partials of dweight / dbias.
static int64_t nhwc_chunks(int N, int C, int HW, int G) {
    NhwcGeom gm;
    if (nhwc_geom("vft_groupnorm_scratch_floats", gm, N, C, HW, G, nullptr, nullptr, nullptr)) return -1;
    return (HW + gm.rows_per_block - 1) / gm.rows_per_block;
}
extern "C" int64_t vft_groupnorm_scratch_floats(int N, int C, int HW, int G, int nhwc, int affine) {
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G != 0) return -1;
    if (!nhwc) return (int64_t)N * C * 2 + (affine ? (int64_t)N * 2 * C : 0);
    const int64_t ch = nhwc_chunks(N, C, HW, G);
    if (ch < 0) return -1;
    return (int64_t)N * G * ch * 2 + (affine ? (int64_t)N * ch * 2 * C : 0);
}

static void finalize_groups(const float* part, float* out, int64_t rows, int nparts, hipStream_t st) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((rows * 2 + 255) / 256)), dim3(256), 0, st, part, out, rows * 2, nparts);
}

// stats: [N*G*2] fp32, written here (saved for backward); scratch: vft_groupnorm_scratch_floats(..., 0, 0) floats
extern "C" int vft_groupnorm_fwd(const void* x, void* y, const void* weight, const void* bias, float* stats, float* scratch,
                                 int N, int C, int HW, int G, float eps, int fuse_silu, void* stream) {
    VFT_CHECK_ARG(x && y && stats && scratch, "vft_groupnorm_fwd: null pointer");
    if (gn_args("vft_groupnorm_fwd", N, C, HW, G)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int V = vec_width(HW, x, y);
#define VFT_GN_FWD(VV)                                                                                                  \
    {                                                                                                                   \
        hipLaunchKernelGGL(gn_stats_kernel<VV>, dim3(N * C), dim3(256), 0, st, (const bf16_t*)x, scratch, C, HW, C / G); \
        finalize_groups(scratch, stats, (int64_t)N * G, C / G, st);                                                     \
        const int64_t nvec = (int64_t)N * C * HW / VV;                                                                  \
        hipLaunchKernelGGL(gn_apply_kernel<VV>, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16_t*)x, stats,        \
                           (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)y, C, HW, C / G, eps, fuse_silu, nvec); \
    }
    switch (V) {
        case 8: VFT_GN_FWD(8); break;
        case 4: VFT_GN_FWD(4); break;
        case 2: VFT_GN_FWD(2); break;
        default: VFT_GN_FWD(1); break;
    }
#undef VFT_GN_FWD
    VFT_CHECK_LAUNCH("vft_groupnorm_fwd");
    return 0;
}

// bstats: [N*G*2] fp32 (written here); dweight/dbias: optional [C] fp32 (written here); scratch:
// vft_groupnorm_scratch_floats(..., 0, dweight || dbias) floats
extern "C" int vft_groupnorm_bwd(const void* x, const void* dy, const float* stats, const void* weight, const void* bias,
                                 void* dx, float* bstats, float* scratch, float* dweight, float* dbias, int N, int C, int HW,
                                 int G, float eps, int fuse_silu, void* stream) {
    VFT_CHECK_ARG(x && dy && stats && dx && bstats && scratch, "vft_groupnorm_bwd: null pointer");
    if (gn_args("vft_groupnorm_bwd", N, C, HW, G)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const bool affine = dweight || dbias;
    float* cpart = affine ? scratch + (int64_t)N * C * 2 : nullptr;
    int V = vec_width(HW, x, dy);
    if (((uintptr_t)dx & (2 * V - 1)) != 0) V = 1;
#define VFT_GN_BWD(VV)                                                                                                    \
    {                                                                                                                     \
        hipLaunchKernelGGL(gn_bwd_stats_kernel<VV>, dim3(N * C), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy,   \
                           stats, (const bf16_t*)weight, (const bf16_t*)bias, scratch, cpart, C, HW, C / G, eps,          \
                           fuse_silu);                                                                                    \
        finalize_groups(scratch, bstats, (int64_t)N * G, C / G, st);                                                      \
        const int64_t nvec = (int64_t)N * C * HW / VV;                                                                    \
        hipLaunchKernelGGL(gn_bwd_dx_kernel<VV>, dim3(grid_for(nvec)), dim3(256), 0, st, (const bf16_t*)x,                \
                           (const bf16_t*)dy, stats, bstats, (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)dx, C,  \
                           HW, C / G, eps, fuse_silu, nvec);                                                              \
    }
    switch (V) {
        case 8: VFT_GN_BWD(8); break;
        case 4: VFT_GN_BWD(4); break;
        case 2: VFT_GN_BWD(2); break;
        default: VFT_GN_BWD(1); break;
    }
#undef VFT_GN_BWD
    if (affine)
        hipLaunchKernelGGL(gn_finalize_affine_kernel, dim3((C + 31) / 32), dim3(256), 0, st, cpart, dweight, dbias, N, C);
    VFT_CHECK_LAUNCH("vft_groupnorm_bwd");
    return 0;
}

// ---- channels-last (NHWC) entry points: same arguments, x / y / dy / dx are [N][HW][C] ----
extern "C" int vft_groupnorm_nhwc_fwd(const void* x, void* y, const void* weight, const void* bias, float* stats,
                                      float* scratch, int N, int C, int HW, int G, float eps, int fuse_silu, void* stream) {
    VFT_CHECK_ARG(x && y && stats && scratch, "vft_groupnorm_nhwc_fwd: null pointer");
    if (gn_args("vft_groupnorm_nhwc_fwd", N, C, HW, G)) return 1;
    NhwcGeom gm;
    if (nhwc_geom("vft_groupnorm_nhwc_fwd", gm, N, C, HW, G, x, y, nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((HW + gm.rows_per_block - 1) / gm.rows_per_block, N), block(gm.CV * gm.R);
    const size_t lds = (size_t)(2 * gm.R + 2) * C * sizeof(float);
    VFT_CHECK_ARG(lds <= 64 * 1024, "vft_groupnorm_nhwc_fwd: C=%d needs %zu bytes of LDS", C, lds);
    hipLaunchKernelGGL(gn_nhwc_stats_kernel, grid, block, lds, st, (const bf16_t*)x, scratch, gm);
    finalize_groups(scratch, stats, (int64_t)N * G, (int)grid.x, st);
    hipLaunchKernelGGL(gn_nhwc_apply_kernel, grid, block, 0, st, (const bf16_t*)x, stats, (const bf16_t*)weight,
                       (const bf16_t*)bias, (bf16_t*)y, gm, eps, fuse_silu);
    VFT_CHECK_LAUNCH("vft_groupnorm_nhwc_fwd");
    return 0;
}

extern "C" int vft_groupnorm_nhwc_bwd(const void* x, const void* dy, const float* stats, const void* weight,
                                      const void* bias, void* dx, float* bstats, float* scratch, float* dweight,
                                      float* dbias, int N, int C, int HW, int G, float eps, int fuse_silu, void* stream) {
    VFT_CHECK_ARG(x && dy && stats && dx && bstats && scratch, "vft_groupnorm_nhwc_bwd: null pointer");
    if (gn_args("vft_groupnorm_nhwc_bwd", N, C, HW, G)) return 1;
    NhwcGeom gm;
    if (nhwc_geom("vft_groupnorm_nhwc_bwd", gm, N, C, HW, G, x, dy, dx)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((HW + gm.rows_per_block - 1) / gm.rows_per_block, N), block(gm.CV * gm.R);
    const bool affine = dweight || dbias;
    float* cpart = affine ? scratch + (int64_t)N * G * grid.x * 2 : nullptr;
    const size_t lds = (size_t)(2 * gm.R + 3) * C * sizeof(float);
    VFT_CHECK_ARG(lds <= 64 * 1024, "vft_groupnorm_nhwc_bwd: C=%d needs %zu bytes of LDS", C, lds);
    hipLaunchKernelGGL(gn_nhwc_bwd_stats_kernel, grid, block, lds, st, (const bf16_t*)x,
                       (const bf16_t*)dy, stats, (const bf16_t*)weight, (const bf16_t*)bias, scratch, cpart, gm, eps,
                       fuse_silu);
    finalize_groups(scratch, bstats, (int64_t)N * G, (int)grid.x, st);
    if (affine)
        hipLaunchKernelGGL(gn_finalize_affine_kernel, dim3((C + 31) / 32), dim3(256), 0, st, cpart, dweight, dbias,
                           (int)(N * grid.x), C);
    hipLaunchKernelGGL(gn_nhwc_bwd_dx_kernel, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, stats, bstats,
                       (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)dx, gm, eps, fuse_silu);
    VFT_CHECK_LAUNCH("vft_groupnorm_nhwc_bwd");
    return 0;
}
