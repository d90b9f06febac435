// Row normalisation kernels (LayerNorm / RMSNorm over the last dimension), bf16 in/out, fp32
// statistics — gfx950.  One wavefront per row, the row is held in registers between the statistics
// and the apply phase, so every tensor is read once and written once (HBM-bound; algorithmic bytes
// = 2·rows·C·2 forward, 3·rows·C·2 backward).
//
// Replaces the ATen chains behind
//   nn.LayerNorm in SDXL TransformerBlock                      src/models/sdxl/denoiser.py:250-252
//   Fp32LayerNorm + modulate (AdaLN) in AuraFlow               src/models/auraflow/denoiser.py:59-76,392-422
//   nn.RMSNorm (+ (1+scale) modulation) in Lumina2             src/models/lumina2/denoiser.py:280-313
//   FP32LayerNorm in Lumina2 FinalLayer                        src/modules/norm.py:9-17
//   per-head q/k norm + RoPE                                   auraflow/denoiser.py:165-199,
//                                                              lumina2/denoiser.py:96-112,137-143,
//                                                              modules/positional_encoding/rope.py:83-111
#include "vft_common.h"

namespace {

constexpr int KIND_LAYER = 0, KIND_RMS = 1, KIND_NONE = 2;
constexpr int MAXCH = 10;  // 8-element chunks per lane: rows up to 64*8*10 = 5120 columns

struct NormParams {
    const bf16_t* x;      // [rows][C] (row stride ldx)
    bf16_t* y;            // [rows][C] contiguous
    const bf16_t* w;      // [C] or null
    const bf16_t* b;      // [C] or null
    const bf16_t* scale;  // [rows/rpb][C] or null: y *= (1 + scale)
    const bf16_t* shift;  // [rows/rpb][C] or null: y += shift
    float* mean;          // [rows]
    float* rstd;          // [rows]
    const bf16_t* dy;     // backward
    bf16_t* dx;           // backward
    const bf16_t* dres;   // backward, optional [rows][C]: gradient arriving through the residual path, added to dx
    const bf16_t* dres2;  // backward, optional: a second residual-path gradient (the input feeds two skip connections)
    int64_t rows;
    int C;
    int64_t ldx;
    int rpb;  // rows per batch element (for scale/shift indexing)
    int kind;
    float eps;
};

__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}

// NCH = 8-element chunks per lane (compile-time: the row lives in NCH*8 registers, so narrow rows keep the
// occupancy that hides HBM latency)
template <int NCH>
__global__ __launch_bounds__(256) void norm_fwd_kernel(NormParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int C = p.C;
    const bf16_t* xr = p.x + row * p.ldx;
    float v[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < C) {
            load8(xr + col, v[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s1 += v[c][i];
                s2 += v[c][i] * v[c][i];
            }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    float mean = 0.f, rstd;
    if (p.kind == KIND_LAYER) {
        mean = s1 / C;
        rstd = rsqrtf(fmaxf(s2 / C - mean * mean, 0.f) + p.eps);
    } else {
        rstd = rsqrtf(s2 / C + p.eps);
    }
    if (lane == 0) {
        if (p.mean) p.mean[row] = mean;
        if (p.rstd) p.rstd[row] = rstd;
    }
    const int64_t bidx = row / p.rpb;
    bf16_t* yr = p.y + row * (int64_t)C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < C) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (v[c][i] - mean) * rstd;
            if (p.w) {
                float t[8];
                load8(p.w + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] *= t[i];
            }
            if (p.b) {
                float t[8];
                load8(p.b + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += t[i];
            }
            if (p.scale) {
                float t[8];
                load8(p.scale + bidx * C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] *= (1.f + t[i]);
            }
            if (p.shift) {
                float t[8];
                load8(p.shift + bidx * C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += t[i];
            }
            store8(yr + col, o);
        }
    }
}

// dx for y = (xhat*w + b)*(1+scale) + shift.   g = dy*(1+scale)*w ;
//   layer: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) ;  rms: dx = rstd*(g - xhat*mean(g*xhat))
template <int NCH>
__global__ __launch_bounds__(256) void norm_bwd_dx_kernel(NormParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int C = p.C;
    const bf16_t* xr = p.x + row * p.ldx;
    const bf16_t* dyr = p.dy + row * (int64_t)C;
    const float mean = p.kind == KIND_LAYER ? p.mean[row] : 0.f;
    const float rstd = p.rstd[row];
    const int64_t bidx = row / p.rpb;
    float xh[NCH][8], g[NCH][8];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < C) {
            load8(xr + col, xh[c]);
            load8(dyr + col, g[c]);
            if (p.scale) {
                float t[8];
                load8(p.scale + bidx * C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) g[c][i] *= (1.f + t[i]);
            }
            if (p.w) {
                float t[8];
                load8(p.w + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) g[c][i] *= t[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[c][i] = (xh[c][i] - mean) * rstd;
                sg += g[c][i];
                sgx += g[c][i] * xh[c][i];
            }
        }
    }
    sg = wave_sum(sg) / C;
    sgx = wave_sum(sgx) / C;
    if (p.kind != KIND_LAYER) sg = 0.f;
    bf16_t* dxr = p.dx + row * (int64_t)C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < C) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = rstd * (g[c][i] - sg - xh[c][i] * sgx);
            if (p.dres != nullptr) {
                float t[8];
                load8(p.dres + row * (int64_t)C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += t[i];
            }
            if (p.dres2 != nullptr) {
                float t[8];
                load8(p.dres2 + row * (int64_t)C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += t[i];
            }
            store8(dxr + col, o);
        }
    }
}

// Affine-parameter gradients of the row norms (needed only when the norm's weight / bias train: the full fine-tune,
// src/models/sdxl/denoiser.py:250-252 under BASELINE config 2):
//     dw[c] = sum_rows dy * (1 + scale) * xhat        db[c] = sum_rows dy * (1 + scale)
// Two launches, no atomics, fixed summation order (bit-reproducible): workgroup (column block of 256, row slab) keeps
// per-thread fp32 sums of 8 columns over every 8th row of its slab, folds its 8 row lanes through LDS in lane order
// and writes one partial row; the second kernel adds the slabs' partials in slab order.  Reads dy and x once
// (2*rows*C*2 bytes): HBM-bound.
struct NormParamGrad {
    const bf16_t* dy;
    const bf16_t* x;
    const bf16_t* scale;
    const float* mean;
    const float* rstd;
    float* part;  // [slabs][2][C]
    void* dw;
    void* db;
    int64_t rows;
    int C;
    int64_t ldx, lddy;
    int rpb, kind, slab_rows, slabs, out_f32, accumulate;
    int64_t dy_bs, part_bs, out_bs;  // batched column sums (vft_colsum_batched): element strides per blockIdx.z; 0 otherwise
};

__global__ __launch_bounds__(256) void norm_bwd_params_partial_kernel(NormParamGrad p) {
    __shared__ float red[2][8][256];
    const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cc * 8;
    const int64_t r0 = (int64_t)blockIdx.y * p.slab_rows;
    const int64_t r1 = r0 + p.slab_rows < p.rows ? r0 + p.slab_rows : p.rows;
    float aw[8], ab[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) aw[i] = 0.f, ab[i] = 0.f;
    if (col < p.C) {
        auto one = [&](int64_t r) {
            float xv[8], g[8];
            load8(p.dy + (int64_t)blockIdx.z * p.dy_bs + r * p.lddy + col, g);
            if (p.x == nullptr) {  // column sums only (vft_colsum)
#pragma unroll
                for (int i = 0; i < 8; ++i) ab[i] += g[i];
                return;
            }
            load8(p.x + r * p.ldx + col, xv);
            const float mean = p.kind == KIND_LAYER ? p.mean[r] : 0.f, rstd = p.rstd[r];
            if (p.scale != nullptr) {
                float t[8];
                load8(p.scale + (r / p.rpb) * p.C + col, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] *= (1.f + t[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                aw[i] += g[i] * ((xv[i] - mean) * rstd);
                ab[i] += g[i];
            }
        };
        int64_t r = r0 + rl;
        for (; r + 8 < r1; r += 16) {  // two rows in flight per thread
            one(r);
            one(r + 8);
        }
        if (r < r1) one(r);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[0][rl][cc * 8 + i] = aw[i], red[1][rl][cc * 8 + i] = ab[i];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < p.C) {
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) sw += red[0][l][threadIdx.x], sb += red[1][l][threadIdx.x];
        float* o = p.part + (int64_t)blockIdx.z * p.part_bs + (int64_t)blockIdx.y * 2 * p.C;
        o[c] = sw;
        o[p.C + c] = sb;
    }
}

// block = 64 columns x 16 slab lanes; blockIdx.y = 0 (dw) / 1 (db)
__global__ __launch_bounds__(1024) void norm_bwd_params_reduce_kernel(NormParamGrad p) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f;
    if (c < p.C)
        for (int s = sl; s < p.slabs; s += 16) a += p.part[(int64_t)blockIdx.z * p.part_bs + ((int64_t)s * 2 + blockIdx.y) * p.C + c];
    red[sl][cl] = a;
    __syncthreads();
    if (sl == 0 && c < p.C) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][cl];
        void* out = blockIdx.y == 0 ? p.dw : p.db;
        if (out != nullptr) {
            const int64_t oc = (int64_t)blockIdx.z * p.out_bs + c;
            if (p.out_f32) ((float*)out)[oc] = t + (p.accumulate ? ((float*)out)[oc] : 0.f);
            else ((bf16_t*)out)[oc] = (bf16_t)(t + (p.accumulate ? (float)((bf16_t*)out)[oc] : 0.f));
        }
    }
}

inline void norm_param_slabs(int64_t rows, int C, int& slab_rows, int& slabs) {
    const int colblocks = (C + 255) / 256;
    int64_t want = 1024 / colblocks;  // ~1024 workgroups: four per CU
    if (want < 1) want = 1;
    int64_t sr = (rows + want - 1) / want;
    if (sr < 64) sr = 64;
    sr = (sr + 15) / 16 * 16;
    slab_rows = (int)sr;
    slabs = (int)((rows + sr - 1) / sr);
}

// ---------------------------------------------------------------------------------------------
// per-head q/k norm over head_dim D (8 elements per lane, D/8 lanes per row) + interleaved-pair
// rotation.  x element (b,s,h,:) at x + b*sb + s*ss + h*sh ; output (B,S,H,D) contiguous.
// cos/sin: fp32 [B or 1][S][D/2]  (cs_bstride = 0 when shared by the batch), or null.
// ---------------------------------------------------------------------------------------------
struct QkParams {
    const bf16_t* x;
    bf16_t* y;
    const bf16_t* w;   // [D] rms weight or null
    const float* cos;
    const float* sin;
    float* rstd;       // [B*S*H] saved (and mean for layer)
    float* mean;
    const bf16_t* dy;
    bf16_t* dx;        // contiguous (B,S,H,D)
    int B, S, H, D;
    int64_t sb, ss, sh;
    int64_t dsb, dss, dsh;  // backward: strides of dy (elements)
    int64_t cs_bstride;
    int kind;
    float eps;
};

template <int D>
__device__ __forceinline__ float group_sum(float v, int lane) {
    constexpr int G = D / 8;  // lanes per row: 8, 12, 16, 32
    if (G == 8 || G == 16 || G == 32) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    } else {
        const int base = (lane / G) * G;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < G; ++i) t += __shfl(v, base + i, 64);
        return t;
    }
}

template <int D, bool BWD>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(QkParams p) {
    constexpr int G = D / 8, RPW = 64 / G;  // rows per wave
    const int lane = threadIdx.x & 63;
    const int gl = lane % G, gr = lane / G;
    const int64_t nrows = (int64_t)p.B * p.S * p.H;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + gr;
    const bool ok = gr < RPW && row < nrows;
    const int64_t rr = ok ? row : 0;
    const int h = (int)(rr % p.H);
    const int64_t bs = rr / p.H;
    const int s = (int)(bs % p.S);
    const int b = (int)(bs / p.S);
    const bf16_t* xr = p.x + b * p.sb + s * p.ss + h * p.sh + gl * 8;
    float v[8];
    load8(xr, v);
    float cs[4] = {1.f, 1.f, 1.f, 1.f}, sn[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.cos) {
        const int64_t off = b * p.cs_bstride + (int64_t)s * (D / 2) + gl * 4;
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.cos + off);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.sin + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cs[i] = c4[i];
            sn[i] = s4[i];
        }
    }
    if (!BWD) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s1 += v[i];
            s2 += v[i] * v[i];
        }
        s1 = group_sum<D>(s1, lane);
        s2 = group_sum<D>(s2, lane);
        float mean = 0.f, rstd = 1.f;
        if (p.kind == KIND_LAYER) {
            mean = s1 / D;
            rstd = rsqrtf(fmaxf(s2 / D - mean * mean, 0.f) + p.eps);
        } else if (p.kind == KIND_RMS) {
            rstd = rsqrtf(s2 / D + p.eps);
        }
        if (ok && gl == 0) {
            if (p.mean) p.mean[row] = mean;
            if (p.rstd) p.rstd[row] = rstd;
        }
        float n[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) n[i] = (v[i] - mean) * rstd;
        if (p.w) {
            float t[8];
            load8(p.w + gl * 8, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) n[i] *= t[i];
        }
        // the reference rounds the normalised value to the activation dtype before rotating
#pragma unroll
        for (int i = 0; i < 8; ++i) n[i] = (float)(bf16_t)n[i];
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = n[2 * i] * cs[i] - n[2 * i + 1] * sn[i];
            o[2 * i + 1] = n[2 * i] * sn[i] + n[2 * i + 1] * cs[i];
        }
        if (ok) store8(p.y + row * D + gl * 8, o);
    } else {
        // dy -> inverse rotation -> norm backward
        float d[8];
        load8(p.dy + b * p.dsb + s * p.dss + h * p.dsh + gl * 8, d);
        float g[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g[2 * i] = d[2 * i] * cs[i] + d[2 * i + 1] * sn[i];
            g[2 * i + 1] = -d[2 * i] * sn[i] + d[2 * i + 1] * cs[i];
        }
        float o[8];
        if (p.kind == KIND_NONE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = g[i];
        } else {
            if (p.w) {
                float t[8];
                load8(p.w + gl * 8, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] *= t[i];
            }
            const float mean = p.kind == KIND_LAYER ? p.mean[rr] : 0.f;
            const float rstd = p.rstd[rr];
            float sg = 0.f, sgx = 0.f;
            float xh[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[i] = (v[i] - mean) * rstd;
                sg += g[i];
                sgx += g[i] * xh[i];
            }
            sg = group_sum<D>(sg, lane) / D;
            sgx = group_sum<D>(sgx, lane) / D;
            if (p.kind != KIND_LAYER) sg = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = rstd * (g[i] - sg - xh[i] * sgx);
        }
        if (ok) store8(p.dx + row * D + gl * 8, o);
    }
}

template <bool BWD>
int launch_qk(const QkParams& p, hipStream_t st) {
    const int64_t nrows = (int64_t)p.B * p.S * p.H;
#define VFT_QK(DD)                                                                                       \
    {                                                                                                    \
        constexpr int RPW = 64 / (DD / 8);                                                               \
        const int64_t nb = (nrows + 4 * RPW - 1) / (4 * RPW);                                            \
        hipLaunchKernelGGL((qk_norm_rope_kernel<DD, BWD>), dim3((unsigned)nb), dim3(256), 0, st, p);     \
    }
    switch (p.D) {
        case 64: VFT_QK(64); break;
        case 96: VFT_QK(96); break;
        case 128: VFT_QK(128); break;
        default: VFT_QK(256); break;
    }
#undef VFT_QK
    return 0;
}

}  // namespace

extern "C" int vft_norm_fwd(const void* x, void* y, const void* weight, const void* bias, const void* scale,
                            const void* shift, float* mean, float* rstd, int64_t rows, int C, int64_t ldx,
                            int rows_per_batch, int kind, float eps, void* stream) {
    VFT_CHECK_ARG(x && y, "vft_norm_fwd: null pointer");
    VFT_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * MAXCH, "vft_norm_fwd: C=%d must be a multiple of 8, <= %d", C, 64 * 8 * MAXCH);
    VFT_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "vft_norm_fwd: alignment");
    VFT_CHECK_ARG(kind == KIND_LAYER || kind == KIND_RMS, "vft_norm_fwd: kind must be 0 (layer) or 1 (rms)");
    VFT_CHECK_ARG(rows_per_batch > 0, "vft_norm_fwd: rows_per_batch must be positive");
    NormParams p{};
    p.x = (const bf16_t*)x, p.y = (bf16_t*)y, p.w = (const bf16_t*)weight, p.b = (const bf16_t*)bias;
    p.scale = (const bf16_t*)scale, p.shift = (const bf16_t*)shift, p.mean = mean, p.rstd = rstd;
    p.rows = rows, p.C = C, p.ldx = ldx, p.rpb = rows_per_batch, p.kind = kind, p.eps = eps;
    const dim3 grid((unsigned)((rows + 3) / 4));
    switch ((C + 511) / 512) {
#define VFT_NORM_FWD(n) case n: hipLaunchKernelGGL(norm_fwd_kernel<n>, grid, dim3(256), 0, (hipStream_t)stream, p); break
        VFT_NORM_FWD(1); VFT_NORM_FWD(2); VFT_NORM_FWD(3); VFT_NORM_FWD(4); VFT_NORM_FWD(5); VFT_NORM_FWD(6);
        case 7: case 8: hipLaunchKernelGGL(norm_fwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
        default: hipLaunchKernelGGL(norm_fwd_kernel<MAXCH>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
#undef VFT_NORM_FWD
    }
    VFT_CHECK_LAUNCH("vft_norm_fwd");
    return 0;
}

extern "C" int vft_norm_bwd_dx(const void* dy, const void* x, const void* weight, const void* scale, const float* mean,
                               const float* rstd, void* dx, const void* dres, const void* dres2, int64_t rows, int C,
                               int64_t ldx, int rows_per_batch, int kind, void* stream) {
    VFT_CHECK_ARG(dy && x && dx && rstd, "vft_norm_bwd_dx: null pointer");
    VFT_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * MAXCH, "vft_norm_bwd_dx: unsupported C=%d", C);
    VFT_CHECK_ARG(kind == KIND_RMS || mean != nullptr, "vft_norm_bwd_dx: layer norm needs the saved mean");
    NormParams p{};
    p.x = (const bf16_t*)x, p.dy = (const bf16_t*)dy, p.dx = (bf16_t*)dx, p.w = (const bf16_t*)weight;
    p.dres = (const bf16_t*)dres, p.dres2 = (const bf16_t*)dres2;
    VFT_CHECK_ARG((((uintptr_t)dres | (uintptr_t)dres2) & 15) == 0, "vft_norm_bwd_dx: misaligned residual gradient");
    p.scale = (const bf16_t*)scale, p.mean = const_cast<float*>(mean), p.rstd = const_cast<float*>(rstd);
    p.rows = rows, p.C = C, p.ldx = ldx, p.rpb = rows_per_batch, p.kind = kind;
    const dim3 grid((unsigned)((rows + 3) / 4));
    switch ((C + 511) / 512) {
#define VFT_NORM_BWD(n) case n: hipLaunchKernelGGL(norm_bwd_dx_kernel<n>, grid, dim3(256), 0, (hipStream_t)stream, p); break
        VFT_NORM_BWD(1); VFT_NORM_BWD(2); VFT_NORM_BWD(3); VFT_NORM_BWD(4); VFT_NORM_BWD(5); VFT_NORM_BWD(6);
        case 7: case 8: hipLaunchKernelGGL(norm_bwd_dx_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
        default: hipLaunchKernelGGL(norm_bwd_dx_kernel<MAXCH>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
#undef VFT_NORM_BWD
    }
    VFT_CHECK_LAUNCH("vft_norm_bwd_dx");
    return 0;
}

extern "C" int64_t vft_norm_bwd_params_scratch_floats(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    int sr, sl;
    norm_param_slabs(rows, C, sr, sl);
    return (int64_t)sl * 2 * C;
}

extern "C" int vft_norm_bwd_params(const void* dy, const void* x, const void* scale, const float* mean, const float* rstd,
                                   void* dw, void* db, float* scratch, int64_t rows, int C, int64_t ldx,
                                   int rows_per_batch, int kind, int out_f32, int accumulate, void* stream) {
    VFT_CHECK_ARG(dy && x && rstd && scratch && (dw || db), "vft_norm_bwd_params: null pointer");
    VFT_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0, "vft_norm_bwd_params: unsupported C=%d", C);
    VFT_CHECK_ARG(kind == KIND_RMS || mean != nullptr, "vft_norm_bwd_params: layer norm needs the saved mean");
    VFT_CHECK_ARG(scale == nullptr || rows_per_batch > 0, "vft_norm_bwd_params: rows_per_batch");
    NormParamGrad p{};
    p.dy = (const bf16_t*)dy, p.x = (const bf16_t*)x, p.scale = (const bf16_t*)scale, p.mean = mean, p.rstd = rstd;
    p.part = scratch, p.dw = dw, p.db = db, p.rows = rows, p.C = C, p.ldx = ldx, p.rpb = rows_per_batch > 0 ? rows_per_batch : 1;
    p.kind = kind, p.out_f32 = out_f32, p.lddy = C, p.accumulate = accumulate;
    norm_param_slabs(rows, C, p.slab_rows, p.slabs);
    hipLaunchKernelGGL(norm_bwd_params_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)p.slabs), dim3(256), 0,
                       (hipStream_t)stream, p);
    hipLaunchKernelGGL(norm_bwd_params_reduce_kernel, dim3((unsigned)((C + 63) / 64), 2), dim3(1024), 0, (hipStream_t)stream, p);
    VFT_CHECK_LAUNCH("vft_norm_bwd_params");
    return 0;
}

// Column sums of a [rows][C] bf16 matrix (row stride ld): the bias gradient of a trainable projection / convolution
// (sum over tokens or pixels of dy), same two ordered passes as the norm parameter gradients.
extern "C" int vft_colsum(const void* dy, void* out, float* scratch, int64_t rows, int C, int64_t ld, int accumulate,
                          int out_f32, void* stream) {
    VFT_CHECK_ARG(dy && out && scratch, "vft_colsum: null pointer");
    VFT_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && ld >= C, "vft_colsum: unsupported C=%d", C);
    VFT_CHECK_ARG(((uintptr_t)dy & 15) == 0, "vft_colsum: misaligned pointer");
    NormParamGrad p{};
    p.dy = (const bf16_t*)dy, p.part = scratch, p.db = out, p.rows = rows, p.C = C, p.lddy = ld, p.rpb = 1;
    p.out_f32 = out_f32, p.accumulate = accumulate;
    norm_param_slabs(rows, C, p.slab_rows, p.slabs);
    hipLaunchKernelGGL(norm_bwd_params_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)p.slabs), dim3(256), 0,
                       (hipStream_t)stream, p);
    hipLaunchKernelGGL(norm_bwd_params_reduce_kernel, dim3((unsigned)((C + 63) / 64), 2), dim3(1024), 0, (hipStream_t)stream, p);
    VFT_CHECK_LAUNCH("vft_colsum");
    return 0;
}

// out[b][c] = sum over the rows of batch element b of dy[b][row][c] (dy: [B][rows][ld >= C] bf16, dense batches): the gradient of a
// per-sample channel vector that was broadcast over the spatial positions — the ResidualBlock's time-embedding projection
// (src/models/sdxl/denoiser.py:524-551, `h + emb_out[..., None, None]`; autograd: aten::sum over H, W).  One launch pair for the
// whole batch (blockIdx.z); scratch: B x vft_norm_bwd_params_scratch_floats(rows, C) floats.  Ordered partial sums.
extern "C" int vft_colsum_batched(const void* dy, void* out, float* scratch, int B, int64_t rows, int C, int64_t ld, int out_f32,
                                  void* stream) {
    VFT_CHECK_ARG(dy && out && scratch, "vft_colsum_batched: null pointer");
    VFT_CHECK_ARG(B > 0 && B < 65536 && rows > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && ld >= C, "vft_colsum_batched: unsupported B=%d C=%d", B, C);
    VFT_CHECK_ARG(((uintptr_t)dy & 15) == 0, "vft_colsum_batched: misaligned pointer");
    NormParamGrad p{};
    p.dy = (const bf16_t*)dy, p.part = scratch, p.db = out, p.rows = rows, p.C = C, p.lddy = ld, p.rpb = 1;
    p.out_f32 = out_f32, p.accumulate = 0;
    norm_param_slabs(rows, C, p.slab_rows, p.slabs);
    p.dy_bs = rows * ld, p.part_bs = (int64_t)p.slabs * 2 * C, p.out_bs = C;
    hipLaunchKernelGGL(norm_bwd_params_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)p.slabs, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, p);
    hipLaunchKernelGGL(norm_bwd_params_reduce_kernel, dim3((unsigned)((C + 63) / 64), 2, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, p);
    VFT_CHECK_LAUNCH("vft_colsum_batched");
    return 0;
}

static int qk_args(const char* name, int B, int S, int H, int D, const int64_t* xs, int kind) {
    VFT_CHECK_ARG(B > 0 && S > 0 && H > 0, "%s: non-positive dimension", name);
    VFT_CHECK_ARG(D == 64 || D == 96 || D == 128 || D == 256, "%s: head_dim %d unsupported", name, D);
    VFT_CHECK_ARG(xs[0] % 8 == 0 && xs[1] % 8 == 0 && xs[2] % 8 == 0, "%s: strides must be multiples of 8", name);
    VFT_CHECK_ARG(kind >= 0 && kind <= 2, "%s: bad kind", name);
    return 0;
}

extern "C" int vft_qk_norm_rope_fwd(const void* x, void* y, const void* weight, const float* cos, const float* sin,
                                    float* mean, float* rstd, int B, int S, int H, int D, const int64_t* x_strides,
                                    int64_t cs_batch_stride, int kind, float eps, void* stream) {
    VFT_CHECK_ARG(x && y, "vft_qk_norm_rope_fwd: null pointer");
    if (qk_args("vft_qk_norm_rope_fwd", B, S, H, D, x_strides, kind)) return 1;
    QkParams p{};
    p.x = (const bf16_t*)x, p.y = (bf16_t*)y, p.w = (const bf16_t*)weight, p.cos = cos, p.sin = sin;
    p.mean = mean, p.rstd = rstd, p.B = B, p.S = S, p.H = H, p.D = D;
    p.sb = x_strides[0], p.ss = x_strides[1], p.sh = x_strides[2], p.cs_bstride = cs_batch_stride;
    p.kind = kind, p.eps = eps;
    launch_qk<false>(p, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_qk_norm_rope_fwd");
    return 0;
}

extern "C" int vft_qk_norm_rope_bwd(const void* dy, const void* x, const void* weight, const float* cos,
                                    const float* sin, const float* mean, const float* rstd, void* dx, int B, int S,
                                    int H, int D, const int64_t* x_strides, const int64_t* dy_strides,
                                    int64_t cs_batch_stride, int kind, void* stream) {
    VFT_CHECK_ARG(dy && x && dx, "vft_qk_norm_rope_bwd: null pointer");
    VFT_CHECK_ARG(dy_strides == nullptr || (dy_strides[0] % 8 == 0 && dy_strides[1] % 8 == 0 && dy_strides[2] % 8 == 0 &&
                                            ((uintptr_t)dy & 15) == 0),
                  "vft_qk_norm_rope_bwd: dy strides must be multiples of 8 elements");
    if (qk_args("vft_qk_norm_rope_bwd", B, S, H, D, x_strides, kind)) return 1;
    VFT_CHECK_ARG(kind == KIND_NONE || rstd != nullptr, "vft_qk_norm_rope_bwd: saved statistics missing");
    QkParams p{};
    p.x = (const bf16_t*)x, p.dy = (const bf16_t*)dy, p.dx = (bf16_t*)dx, p.w = (const bf16_t*)weight;
    p.cos = cos, p.sin = sin, p.mean = const_cast<float*>(mean), p.rstd = const_cast<float*>(rstd);
    p.B = B, p.S = S, p.H = H, p.D = D;
    p.sb = x_strides[0], p.ss = x_strides[1], p.sh = x_strides[2], p.cs_bstride = cs_batch_stride, p.kind = kind;
    p.dsb = dy_strides ? dy_strides[0] : (int64_t)S * H * D, p.dss = dy_strides ? dy_strides[1] : (int64_t)H * D;
    p.dsh = dy_strides ? dy_strides[2] : D;
    launch_qk<true>(p, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_qk_norm_rope_bwd");
    return 0;
}
