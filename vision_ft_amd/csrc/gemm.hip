// bf16 "NT" GEMM on MFMA for gfx950:  C[M,N] = A[M,K] · B[N,K]ᵀ (+ bias[N]),  fp32 accumulate.
// B (the weight, row-major [out, in] exactly as nn.Linear stores it) is either bf16 or **NF4** in
// the bitsandbytes layout, dequantised in the staging prologue — the packed 4-bit codes and the
// per-64 absmax are what travels from HBM; bf16 weights exist only as LDS tiles.
//
// Replaces, for the linear projections of the denoisers:
//   nn.Linear -> cuBLAS                                   e.g. src/models/lumina2/denoiser.py:123-131
//   bnb.nn.Linear4bit.forward -> bnb.matmul_4bit           src/modules/quant/bnb.py:37-129
//     (kDequantizeBlockwise + GEMM of bitsandbytes 0.48.2; NF4 code book, blocksize 64,
//      optional double-quantised absmax: absmax = code2[q8] * absmax2[blk/256] + offset)
// The backward data gradient dX[M,K] = dY[M,N] · W[N,K] runs the same kernel on the transposed
// weight (for NF4: the `TRANS` staging variant dequantises W[n0.., k0..] and stores it transposed).
//
// Tiling: 128x128 output tile per 256-thread workgroup, BK = 64; 4 waves as 2x2, each 64x64 =
// 2x2 v_mfma_f32_32x32x16_bf16 tiles.  The MFMA "A" operand is the weight tile and the "B" operand
// the activation tile, so an accumulator register group holds 4 consecutive output features of one
// token -> 8-byte epilogue stores and a per-register bias.  Global->LDS staging goes through
// registers one K-tile ahead (loads issued before the MFMAs of the current tile); LDS rows are
// 128 B with an XOR swizzle of the 16-byte chunk index.
#include "vft_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

struct GemmParams {
    const bf16_t* a;      // [M][lda]
    const bf16_t* b;      // bf16 weight [N][ldb]   (plain variant)
    bf16_t* c;            // [M][ldc]
    const bf16_t* bias;   // [N] or null
    int M, N, K;
    int64_t lda, ldb, ldc;
    // NF4 weight (logical shape [QN][QK] row-major, blocksize 64 along the flattened index)
    const uint8_t* q_packed;    // QN*QK/2 bytes, element 2i in the high nibble
    const uint8_t* q_absmax8;   // [nblocks] (double quant) or null
    const float* q_absmax32;    // [nblocks] fp32 absmax (no double quant) or null
    const float* q_absmax2;     // [ceil(nblocks/256)]
    const float* q_code2;       // [256]
    const float* q_code;        // [16]
    float q_offset;
    int QN, QK;
};

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// element offset of 16-byte chunk `ch` (0..7) of row `row` in a [rows][64] bf16 tile
__device__ __forceinline__ int swz(int row, int ch) { return row * BK + ((ch ^ (row & 7)) << 3); }

__device__ __forceinline__ float nf4_absmax(const GemmParams& p, int64_t blk) {
    if (p.q_absmax8) return p.q_code2[p.q_absmax8[blk]] * p.q_absmax2[blk >> 8] + p.q_offset;
    return p.q_absmax32[blk];
}

// MODE 0: bf16 weight.  MODE 1: NF4 weight used as stored (forward: B[n][k] = W[n][k]).
// MODE 2: NF4 weight used transposed (dgrad: B[n'][k'] = W[k'][n'], i.e. N = QK, K = QN).
template <int MODE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t As[BM * BK];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[BN * BK];
    __shared__ float code_s[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    // tile mapping: consecutive workgroups walk M first (they share the weight tile through L2)
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    if (MODE != 0 && tid < 16) code_s[tid] = p.q_code[tid];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- staging registers -------------------------------------------------------------------
    u32x4 ra[4], rb[4];
    float rabs[2] = {0.f, 0.f};
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 256 + tid;
            const int row = c >> 3, ch = c & 7;
            const int gm = min(m0 + row, p.M - 1);
            ra[q] = *reinterpret_cast<const u32x4*>(p.a + (int64_t)gm * p.lda + k0 + ch * 8);
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 256 + tid;
                const int row = c >> 3, ch = c & 7;
                const int gn = min(n0 + row, p.N - 1);
                rb[q] = *reinterpret_cast<const u32x4*>(p.b + (int64_t)gn * p.ldb + k0 + ch * 8);
            }
        } else if (MODE == 1) {
            // thread -> (weight row n, 32-element half of the 64-wide k tile); one quant block per row/tile
            const int row = tid >> 1, half = tid & 1;
            const int gn = min(n0 + row, p.N - 1);
            const int64_t e0 = (int64_t)gn * p.QK + k0;
            rb[0] = *reinterpret_cast<const u32x4*>(p.q_packed + (e0 >> 1) + half * 16);
            rabs[0] = nf4_absmax(p, e0 >> 6);
        } else {
            // transposed use: this tile needs W[k0 .. k0+64)[n0 .. n0+128): thread -> (W row = k index,
            // 32-element segment of the 128 n's); two segments per thread -> 64 k-rows x 4 segments
#pragma unroll
            for (int q = 0; q < 1; ++q) {
                const int wrow = tid >> 2, seg = tid & 3;  // wrow 0..63, seg 0..3 (32 elements each)
                const int gk = k0 + wrow;                  // < K == QN by construction (K % 64 == 0)
                const int gn = min(n0 + seg * 32, p.QK - 32);
                const int64_t e0 = (int64_t)gk * p.QK + gn;
                rb[0] = *reinterpret_cast<const u32x4*>(p.q_packed + (e0 >> 1));
                rabs[0] = nf4_absmax(p, e0 >> 6);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 256 + tid;
            *reinterpret_cast<u32x4*>(As + swz(c >> 3, c & 7)) = ra[q];
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 256 + tid;
                *reinterpret_cast<u32x4*>(Bs + swz(c >> 3, c & 7)) = rb[q];
            }
        } else if (MODE == 1) {
            const int row = tid >> 1, half = tid & 1;
            const float am = rabs[0];
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // each dword = 4 bytes = 8 weights = one 16-byte bf16 chunk
                const unsigned int d = rb[0][w];
                bf16x8 v;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const unsigned int byte = (d >> (8 * bb)) & 0xffu;
                    v[2 * bb] = (bf16_t)(code_s[byte >> 4] * am);
                    v[2 * bb + 1] = (bf16_t)(code_s[byte & 15] * am);
                }
                *reinterpret_cast<bf16x8*>(Bs + swz(row, half * 4 + w)) = v;
            }
        } else {
            // scatter-transpose: W[k][n..n+32) -> Bs[n_local][k_local]
            const int wrow = tid >> 2, seg = tid & 3;
            const float am = rabs[0];
            const int gn_base = min(n0 + seg * 32, p.QK - 32) - n0;  // may be negative-shifted at the N tail
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned int d = rb[0][w];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const unsigned int byte = (d >> (8 * bb)) & 0xffu;
                    const int nl = gn_base + w * 8 + bb * 2;
                    if (nl >= 0 && nl < BN) Bs[swz(nl, wrow >> 3) + (wrow & 7)] = (bf16_t)(code_s[byte >> 4] * am);
                    if (nl + 1 >= 0 && nl + 1 < BN) Bs[swz(nl + 1, wrow >> 3) + (wrow & 7)] = (bf16_t)(code_s[byte & 15] * am);
                }
            }
        }
    };

    const int nkt = p.K / BK;
    fetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        stash();
        __syncthreads();
        if (kt + 1 < nkt) fetch((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fw[2], fx[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fw[i] = *reinterpret_cast<const bf16x8*>(Bs + swz(wn * 64 + i * 32 + r, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fx[j] = *reinterpret_cast<const bf16x8*>(As + swz(wm * 64 + j * 32 + r, ks * 2 + h));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
        }
    }

    // ---- epilogue: acc[i][j] rows (regs) = output feature n, col (lane) = token m -------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + r;
        if (m >= p.M) continue;
        bf16_t* crow = p.c + (int64_t)m * p.ldc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * h;
                if (n + 3 < p.N) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * g + e];
                    if (p.bias) {
                        const bf16x4 bv = *reinterpret_cast<const bf16x4*>(p.bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += (float)bv[e];
                    }
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)o[e];
                    *reinterpret_cast<bf16x4*>(crow + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) crow[n + e] = (bf16_t)(acc[i][j][4 * g + e] + (p.bias ? (float)p.bias[n + e] : 0.f));
                }
            }
    }
}

// NF4 -> bf16 weight materialisation (used for the weight-gradient-free paths that still want a
// dense copy, and by tests):  out[QN][QK]
__global__ __launch_bounds__(256) void nf4_dequant_kernel(GemmParams p, bf16_t* __restrict__ out) {
    const int64_t nvec = (int64_t)p.QN * p.QK / 8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        const unsigned int d = *reinterpret_cast<const unsigned int*>(p.q_packed + e * 4);
        const float am = nf4_absmax(p, (e * 8) >> 6);
        bf16x8 v;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const unsigned int byte = (d >> (8 * bb)) & 0xffu;
            v[2 * bb] = (bf16_t)(p.q_code[byte >> 4] * am);
            v[2 * bb + 1] = (bf16_t)(p.q_code[byte & 15] * am);
        }
        *reinterpret_cast<bf16x8*>(out + e * 8) = v;
    }
}

int common_checks(const char* name, const void* a, const void* c, int M, int N, int K, int64_t lda, int64_t ldc) {
    VFT_CHECK_ARG(a && c, "%s: null pointer", name);
    VFT_CHECK_ARG(M > 0 && N > 0 && K > 0, "%s: non-positive dimension", name);
    VFT_CHECK_ARG(K % BK == 0, "%s: K=%d must be a multiple of %d", name, K, BK);
    VFT_CHECK_ARG(N % 4 == 0 && lda % 8 == 0 && ldc % 4 == 0, "%s: N %% 4, lda %% 8, ldc %% 4 must be 0", name);
    VFT_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)c & 7) == 0, "%s: misaligned activation pointer", name);
    return 0;
}

unsigned grid_of(int M, int N) { return (unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN)); }

}  // namespace

extern "C" int vft_gemm_nt_bf16(const void* a, const void* b, void* c, const void* bias, int M, int N, int K, int64_t lda,
                                int64_t ldb, int64_t ldc, void* stream) {
    if (common_checks("vft_gemm_nt_bf16", a, c, M, N, K, lda, ldc)) return 1;
    VFT_CHECK_ARG(b != nullptr && ldb % 8 == 0 && ((uintptr_t)b & 15) == 0, "vft_gemm_nt_bf16: bad weight pointer / stride");
    GemmParams p{};
    p.a = (const bf16_t*)a, p.b = (const bf16_t*)b, p.c = (bf16_t*)c, p.bias = (const bf16_t*)bias;
    p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldb = ldb, p.ldc = ldc;
    hipLaunchKernelGGL(gemm_nt_kernel<0>, dim3(grid_of(M, N)), dim3(256), 0, (hipStream_t)stream, p);
    VFT_CHECK_LAUNCH("vft_gemm_nt_bf16");
    return 0;
}

static int fill_nf4(GemmParams& p, const char* name, const void* packed, const void* absmax, int absmax_is_u8,
                    const float* absmax2, const float* code2, const float* code, float offset, int QN, int QK) {
    VFT_CHECK_ARG(packed && absmax && code, "%s: null quantisation state", name);
    VFT_CHECK_ARG(!absmax_is_u8 || (absmax2 && code2), "%s: double-quantised absmax needs nested_absmax and nested_quant_map", name);
    VFT_CHECK_ARG(QN > 0 && QK > 0 && QK % 64 == 0 && ((uintptr_t)packed & 15) == 0, "%s: weight must have in_features %% 64 == 0 and a 16-byte aligned payload", name);
    p.q_packed = (const uint8_t*)packed;
    p.q_absmax8 = absmax_is_u8 ? (const uint8_t*)absmax : nullptr;
    p.q_absmax32 = absmax_is_u8 ? nullptr : (const float*)absmax;
    p.q_absmax2 = absmax2, p.q_code2 = code2, p.q_code = code, p.q_offset = offset, p.QN = QN, p.QK = QK;
    return 0;
}

// y[M, out] = x[M, in] · W_nf4[out, in]ᵀ (+ bias)        (transpose_w = 0)
// dx[M, in] = dy[M, out] · W_nf4[out, in]                 (transpose_w = 1)
extern "C" int vft_gemm_nf4(const void* a, void* c, const void* bias, const void* packed, const void* absmax,
                            int absmax_is_u8, const float* nested_absmax, const float* nested_code, const float* code,
                            float nested_offset, int out_features, int in_features, int M, int64_t lda, int64_t ldc,
                            int transpose_w, void* stream) {
    const int N = transpose_w ? in_features : out_features;
    const int K = transpose_w ? out_features : in_features;
    if (common_checks("vft_gemm_nf4", a, c, M, N, K, lda, ldc)) return 1;
    GemmParams p{};
    if (fill_nf4(p, "vft_gemm_nf4", packed, absmax, absmax_is_u8, nested_absmax, nested_code, code, nested_offset,
                 out_features, in_features))
        return 1;
    VFT_CHECK_ARG(!transpose_w || in_features % 32 == 0, "vft_gemm_nf4: transposed use needs in_features %% 32 == 0");
    p.a = (const bf16_t*)a, p.c = (bf16_t*)c, p.bias = (const bf16_t*)bias;
    p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldc = ldc;
    if (transpose_w)
        hipLaunchKernelGGL(gemm_nt_kernel<2>, dim3(grid_of(M, N)), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<1>, dim3(grid_of(M, N)), dim3(256), 0, (hipStream_t)stream, p);
    VFT_CHECK_LAUNCH("vft_gemm_nf4");
    return 0;
}

extern "C" int vft_nf4_dequant(void* out, const void* packed, const void* absmax, int absmax_is_u8,
                               const float* nested_absmax, const float* nested_code, const float* code,
                               float nested_offset, int out_features, int in_features, void* stream) {
    VFT_CHECK_ARG(out != nullptr && ((uintptr_t)out & 15) == 0, "vft_nf4_dequant: bad output pointer");
    GemmParams p{};
    if (fill_nf4(p, "vft_nf4_dequant", packed, absmax, absmax_is_u8, nested_absmax, nested_code, code, nested_offset,
                 out_features, in_features))
        return 1;
    const int64_t nvec = (int64_t)out_features * in_features / 8;
    int64_t nb = (nvec + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(nf4_dequant_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, (bf16_t*)out);
    VFT_CHECK_LAUNCH("vft_nf4_dequant");
    return 0;
}
