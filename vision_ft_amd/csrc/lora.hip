// Low-rank adapter kernels (LoRA), bf16 in/out, fp32 accumulate — gfx950.
//
// Replace the three extra cuBLAS launches + scale + add that the reference's LoRALinear issues per
// layer (src/modules/peft/lora.py:92-104:  linear(x) + lora_up(lora_down(x)) * (alpha/rank)) and the
// four skinny GEMMs autograd adds in backward.  Every kernel is an HBM-bound streaming pass over a
// token-major activation; the rank-sized operands come from L2 / LDS / registers.
//
//   vft_lora_fwd     y[M,N] += s · (x[M,K] · Aᵀ) · Bᵀ      one launch: per 16-row block, phase 1 builds
//                                                           t = x·Aᵀ (MFMA 16x16x32, K split over the 4
//                                                           waves, reduced through LDS), phase 2 streams
//                                                           the block's rows of y.  Also emits t (saved
//                                                           for backward) and Bᵀ (so backward's dy·B is
//                                                           again a k-contiguous product).
//   vft_lora_bwd_dx  dx[M,K] += s · (dy[M,N] · B) · A       same structure; emits u = dy·B.
//   vft_lora_down    t[M,r]  = x[M,K] · Aᵀ                  phase 1 alone, for consumers that apply the update
//                                                           themselves (GEGLU kernels, GEMM rank extension).
//   vft_lora_wgrad2  dA = s · uᵀ x,  dB = s · dyᵀ t         both weight gradients in one launch pair
//                                                           (partials per 256-row block, then a
//                                                           deterministic reduce).
//
// Algorithmic HBM bytes per layer (bf16): fwd  2·M·K + 4·M·N ; bwd_dx  2·M·N + 4·M·K ;
// wgrad2  2·M·K + 2·M·N.
#include "vft_common.h"

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
    return z;
}

// ---------------------------------------------------------------------------------------------
// One block = 16 token rows, 256 threads.
//   phase 1:  p[16][r] = X[16 rows][K] · Wᵀ,  W = w1[r][K] row-major (k contiguous)
//   phase 2:  Y[16 rows][N] += s · p · W2,    W2(j, n):  NJ layout  w2[n*r + j]   (fwd: lora_up (N,r))
//                                                        JN layout  w2[j*N + n]   (bwd: lora_down (r,K))
// RC = 4 / 8 / 16 >= r: the rank padded to the register tile; phase 2 keeps coefficients and activations as bf16
// pairs (v_dot2c_f32_bf16).  N == 0 (vft_lora_down): phase 1 only.
// ---------------------------------------------------------------------------------------------
template <int RC, bool NJ>
__global__ __launch_bounds__(256) void lora_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                        bf16_t* __restrict__ p_out, bf16_t* __restrict__ w2t_out,
                                                        const bf16_t* __restrict__ res, int64_t ldres,
                                                        int M, int K, int N, int r, int64_t ldx, int64_t ldy, float s) {
    __shared__ float red[4][16][17];
    __shared__ float ps[16][17];
    __shared__ bf16x2 pt[16][9];  // the same activations as bf16 pairs
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.x * 16;
    // ---- phase 1 ----
    {
        const int row = m0 + (lane & 15);
        const int q = lane >> 4;
        const int j = lane & 15;
        const bool rok = row < M, jok = j < r;
        const bf16_t* xr = x + (int64_t)row * ldx;
        const bf16_t* wr = w1 + (int64_t)j * K;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int nks = (K + 31) / 32;
#pragma unroll 4
        for (int ks = wv; ks < nks; ks += 4) {
            const int k = ks * 32 + 8 * q;
            bf16x8 a = zero8(), b = zero8();
            if (k < K) {
                if (rok) a = *reinterpret_cast<const bf16x8*>(xr + k);
                if (jok) b = *reinterpret_cast<const bf16x8*>(wr + k);
            }
            acc = mfma16(a, b, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wv][q * 4 + i][j] = acc[i];
    }
    __syncthreads();
    {
        const int row = tid >> 4, j = tid & 15;
        const float v = red[0][row][j] + red[1][row][j] + red[2][row][j] + red[3][row][j];
        ps[row][j] = v;
        {  // rank-r activations as bf16 pairs for phase 2 (they are a bf16 tensor in the reference)
            const float vn = __shfl_down(v, 1, 64);
            if ((j & 1) == 0) {
                bf16x2 pr;
                pr[0] = (bf16_t)v;
                pr[1] = (bf16_t)vn;
                pt[row][j >> 1] = pr;
            }
        }
        // (vft_lora_down, N == 0: the activations leave pre-scaled, so that consumers add them with the adapter's alpha / rank
        // already applied)
        if (p_out != nullptr && j < r && m0 + row < M) p_out[(int64_t)(m0 + row) * r + j] = (bf16_t)(N == 0 ? v * s : v);
    }
    __syncthreads();
    if (N == 0) return;  // vft_lora_down: only the rank-r activations are wanted
    // ---- phase 2 ----
    // Thread map: the N/8 column groups are cut into passes of cw <= 128 groups and the 256 threads
    // cover cw groups x rp rows at once, so narrow layers (N = 640: 80 groups) still use ~all lanes.
    const int nrows = min(16, M - m0);
    const int G = N >> 3;
    const int npass = (G + 127) >> 7;
    const int cw = (G + npass - 1) / npass;
    const int rp = 256 / cw;
    const int lr = tid / cw, lc = tid - lr * cw;
    if (lr >= rp) return;
    {
        // Coefficients and activations as bf16 pairs, two MACs per v_dot2c_f32_bf16: half the VALU work and half the
        // coefficient registers of an fp32-FMA form (whose 128 coefficient registers at rank 16 left two waves per
        // SIMD to hide the HBM latency of a read-modify-write stream).
        constexpr int HP = RC / 2;
        for (int c0 = 0; c0 < G; c0 += cw) {
            if (c0 + lc >= G) break;
            const int n0 = (c0 + lc) * 8;
            bf16x2 wpk[8][HP];
            if (NJ) {
                if (r == RC) {  // the 8 columns' coefficients are one contiguous run of 8*r elements: RC 16-byte loads
                    bf16x8 wv[RC];
#pragma unroll
                    for (int q = 0; q < RC; ++q) wv[q] = *reinterpret_cast<const bf16x8*>(w2 + (int64_t)n0 * RC + 8 * q);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int jj = 0; jj < HP; ++jj) {
                            bf16x2 t2;
                            t2[0] = wv[(c * RC + 2 * jj) >> 3][(c * RC + 2 * jj) & 7];
                            t2[1] = wv[(c * RC + 2 * jj + 1) >> 3][(c * RC + 2 * jj + 1) & 7];
                            wpk[c][jj] = t2;
                        }
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int jj = 0; jj < HP; ++jj) {
                            bf16x2 t2;
                            t2[0] = (2 * jj < r) ? w2[(int64_t)(n0 + c) * r + 2 * jj] : (bf16_t)0.f;
                            t2[1] = (2 * jj + 1 < r) ? w2[(int64_t)(n0 + c) * r + 2 * jj + 1] : (bf16_t)0.f;
                            wpk[c][jj] = t2;
                        }
                }
                if (w2t_out != nullptr && blockIdx.x == 0 && lr == 0) {
#pragma unroll
                    for (int j = 0; j < RC; ++j)
                        if (j < r) {
                            bf16x8 v;
#pragma unroll
                            for (int c = 0; c < 8; ++c) v[c] = wpk[c][j >> 1][j & 1];
                            *reinterpret_cast<bf16x8*>(w2t_out + (int64_t)j * N + n0) = v;
                        }
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < HP; ++jj) {
                    bf16x8 va = zero8(), vb = zero8();
                    if (2 * jj < r) va = *reinterpret_cast<const bf16x8*>(w2 + (int64_t)(2 * jj) * N + n0);
                    if (2 * jj + 1 < r) vb = *reinterpret_cast<const bf16x8*>(w2 + (int64_t)(2 * jj + 1) * N + n0);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        bf16x2 t2;
                        t2[0] = va[c];
                        t2[1] = vb[c];
                        wpk[c][jj] = t2;
                    }
                }
            }
            for (int i0 = lr; i0 < nrows; i0 += 4 * rp) {
                bf16x8 v[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = i0 + b * rp;
                    if (i < nrows) v[b] = *reinterpret_cast<const bf16x8*>(y + (int64_t)(m0 + i) * ldy + n0);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = i0 + b * rp;
                    if (i >= nrows) break;
                    bf16x2 tp[HP];
#pragma unroll
                    for (int jj = 0; jj < HP; ++jj) tp[jj] = pt[i][jj];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float d = 0.f;
#pragma unroll
                        for (int jj = 0; jj < HP; ++jj) d = __builtin_amdgcn_fdot2_f32_bf16(wpk[c][jj], tp[jj], d, false);
                        v[b][c] = (bf16_t)fmaf(s, d, (float)v[b][c]);
                    }
                    if (res != nullptr) {
                        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(res + (int64_t)(m0 + i) * ldres + n0);
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[b][c] = (bf16_t)((float)v[b][c] + (float)rv[c]);
                    }
                    *reinterpret_cast<bf16x8*>(y + (int64_t)(m0 + i) * ldy + n0) = v[b];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// vft_lora_down proper: t[M][r] = s · x[M][K] · w1[r][K]ᵀ with 32 token rows per workgroup.  Same MFMA arrangement as
// phase 1 of lora_rows_kernel (16x16x32, K split over the 4 waves, partial sums reduced through LDS), but every w1
// fragment a wave loads serves TWO 16-row groups: at rank 16 — where all 64 lanes load a w1 fragment per k-step and the
// kernel is bound by the texture-address path, not by HBM — that is 3 loads per 2 MFMAs instead of 2 per 1.
// ---------------------------------------------------------------------------------------------
template <int RG>  // 16-row groups per workgroup
__global__ __launch_bounds__(256) void lora_down32_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w1,
                                                          bf16_t* __restrict__ t_out, int M, int K, int r, int64_t ldx,
                                                          float s) {
    __shared__ float red[4][16 * RG][17];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.x * 16 * RG;
    {
        const int q = lane >> 4, j = lane & 15;
        const bool jok = j < r;
        const bf16_t* xr[RG];
        bool ok[RG];
        f32x4 acc[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int row = m0 + 16 * g + j;
            ok[g] = row < M;
            xr[g] = x + (int64_t)row * ldx;
            acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const bf16_t* wr = w1 + (int64_t)j * K;
        const int nks = (K + 31) / 32;
#pragma unroll 4
        for (int ks = wv; ks < nks; ks += 4) {
            const int k = ks * 32 + 8 * q;
            bf16x8 a[RG], b = zero8();
#pragma unroll
            for (int g = 0; g < RG; ++g) a[g] = zero8();
            if (k < K) {
#pragma unroll
                for (int g = 0; g < RG; ++g)
                    if (ok[g]) a[g] = *reinterpret_cast<const bf16x8*>(xr[g] + k);
                if (jok) b = *reinterpret_cast<const bf16x8*>(wr + k);
            }
#pragma unroll
            for (int g = 0; g < RG; ++g) acc[g] = mfma16(a[g], b, acc[g]);
        }
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wv][16 * g + q * 4 + i][j] = acc[g][i];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < RG; ++h) {
        const int row = h * 16 + (tid >> 4), j = tid & 15;
        const float v = red[0][row][j] + red[1][row][j] + red[2][row][j] + red[3][row][j];
        if (j < r && m0 + row < M) t_out[(int64_t)(m0 + row) * r + j] = (bf16_t)(v * s);
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad stage 1: P[mb][k][j] = Σ_{m in 256-row block mb} X[m][k] · T[m][j]     (fp32, r <= 32)
// Gᵀ tile (32 k x 32 j) per MFMA: A = Xᵀ via ds_read_b64_tr_b16 of the row-major staged X tile,
// B = T rows (staged in LDS) in the accumulator-k order.  block = 4 waves; 64-row slabs, each wave
// owns 32 of the tile's 128 columns; the next slab is fetched into registers under the MFMAs.
// blockIdx.z selects the problem (0: X=x,T=u -> dA ; 1: X=dy,T=t -> dB).
// ---------------------------------------------------------------------------------------------
constexpr int WG_COLS = 128, WG_LD = WG_COLS + 8;
constexpr int UCT = 1;  // column tiles per workgroup of the dy pass that also produces u (see lora_wgrad_partial_kernel)

// rows of X each workgroup reduces: large enough that few partials are written, small enough that
// (row blocks x column tiles x 2 problems) still fills 256 CUs a few times over
static int g_wgrad_target_wgs = 0;  // A/B (vft_lora_wgrad_target): workgroups the row blocking aims at; 0 = automatic
__host__ inline int wgrad_rows_per_block(int M, int K, int N) {
    const int64_t ctiles = (int64_t)(K + WG_COLS - 1) / WG_COLS + (N + WG_COLS - 1) / WG_COLS;
    // measured on MI355X: ~2048 workgroups (8 per CU in flight) are 4-10 % faster than ~1024 once the two operands hold
    // 240 MB or more; below that the doubled partial sums cost more in the reduce than the deeper queue gains
    const int target = g_wgrad_target_wgs > 0 ? g_wgrad_target_wgs : (2.0 * M * (K + N) >= 240e6 ? 2048 : 1024);
    int64_t rpb = ((int64_t)M * ctiles / target + 63) / 64 * 64;
    if (rpb < 256) rpb = 256;
    if (rpb > 4096) rpb = 4096;
    return (int)rpb;
}

struct WgradProblem {
    const bf16_t* x;
    const bf16_t* t;
    float* part;
    bf16_t* out[3];  // one gradient tensor per adapter group (1, or 3 for a fused q|k|v), each dense
    int K;           // columns of x: the long dimension of this problem's gradient
    int64_t ldx;
    int kind;        // 0: d(lora_down) (rsub, K) per group, group = j / rsub;  1: d(lora_up) (K / groups, rsub) per group,
                     // group = k / ksub and only the diagonal block j in [g rsub, (g+1) rsub) exists
    int groups, rsub, ksub;
};

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int ld, int row0, int col0, int lane) {
    const int h = lane >> 5;
    const int i = lane & 15;
    const bf16_t* a0 = tile + (row0 + 4 * h + (i >> 2)) * ld + col0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * ld));
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// One deferred reduction (vft_lora_wgrad_reduce_many): written into a device table BY the partial kernel itself (one lane
// of its first workgroup per problem), so that queueing a layer's reduction costs no host-to-device copy and is
// capturable into a HIP graph.
struct ReduceDesc {
    WgradProblem p;
    int nmb, r, accumulate;
    float scale;
};

// Round 5 — the data gradient's low-rank activations u = s * dy * B ride in the pass over dy (`up_t` / `u_part` non-null, the
// problem with X = dy): the workgroup's 64 x 128 slab of dy is in LDS anyway, so four v_mfma_f32_16x16x32_bf16 per wave and
// slab give its contribution to u for those 64 rows from these 128 columns — u_part[column tile][row][rank] (fp32), summed
// over the column tiles in order by lora_u_reduce_kernel.  That replaces vft_lora_down(dy): one full read of dy per adapted
// layer and backward pass (the FeedForward up-projection's dy is 1 GB at the SDXL headline shape) for a partial-sum round
// trip of 6 % of it at rank 4.
// CT = column tiles (of 128) one workgroup walks per row slab.  1 everywhere.  CT = 4 for the pass over dy that also produces
// u (partial sums per 512-column strip: a quarter of the u_part round trip) was built and measured in round 5: SLOWER — the
// grid shrinks fourfold against a 256-row floor on the row blocks (288 workgroups at M 24 576, N 1 280) and the pass loses
// more to the thinner stream than the smaller round trip returns (SDXL-LoRA step 462.7 ms against 454.6 with CT = 1 and
// 459.2 with the two-kernel form of round 4, one box).  UCT below selects it.
template <int CT>
__global__ __launch_bounds__(256) void lora_wgrad_partial_kernel(WgradProblem p0, WgradProblem p1, int M, int r,
                                                                 int rows_per_block, ReduceDesc* slot, int nmb_total,
                                                                 float scale, int accumulate, int zoff,
                                                                 const bf16_t* __restrict__ up_t, float* __restrict__ u_part) {
    const int pz = blockIdx.z + zoff;  // which problem: 0 (X = x, T = u -> d lora_down) or 1 (X = dy, T = t -> d lora_up)
    if (slot != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        ReduceDesc d;
        d.p = pz == 0 ? p0 : p1;
        d.nmb = nmb_total, d.r = r, d.accumulate = accumulate, d.scale = scale;
        slot[blockIdx.z] = d;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);                 // [64][WG_LD]
    bf16_t* Ts = reinterpret_cast<bf16_t*>(smem + 64 * WG_LD * 2);  // [64][r]
    const WgradProblem& p = pz == 0 ? p0 : p1;
    const int K = p.K;
    const int k0 = blockIdx.x * (CT * WG_COLS);
    if (k0 >= K) return;  // block-uniform
    const int nct = min(CT, (K - k0 + WG_COLS - 1) / WG_COLS);  // column tiles of this strip that exist (block-uniform)
    const bf16_t* __restrict__ x = p.x;
    const bf16_t* __restrict__ t = p.t;
    const int64_t ldx = p.ldx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5, j = lane & 31;
    const int mb = blockIdx.y;
    const int mstart = mb * rows_per_block;
    const int mend = min(M, mstart + rows_per_block);
    const int tvecs = 64 * r / 8;  // 16-byte vectors in one 64-row slab of T
    f32x16 acc[CT];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cc][i] = 0.f;

    // u path (problem 1 only): this lane's lora_up^T fragments for every 32-column k-step of the strip — A operand of the
    // 16x16x32 MFMA: rank row lane & 15, columns k0 + 128 cc + 32 ks + 8 (lane >> 4) .. + 7 (rows >= r, columns >= K: zeros)
    const bool with_u = pz == 1 && u_part != nullptr;
    bf16x8 upf[CT][4];
    if (with_u) {
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int col = k0 + cc * WG_COLS + ks * 32 + 8 * (lane >> 4);
                upf[cc][ks] = zero8();
                if ((lane & 15) < r && col < K) upf[cc][ks] = *reinterpret_cast<const bf16x8*>(up_t + (int64_t)(lane & 15) * K + col);
            }
    }
    u32x4 xr[4];
    bf16x8 tr8 = zero8();
    auto fetch = [&](int r0, int cc) {  // the X tile of (slab r0, column tile cc); with cc == 0 also the slab's rows of T
        const int kc = k0 + cc * WG_COLS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 256 + tid;
            const int row = c / 16, col = (c % 16) * 8;
            xr[q] = u32x4{0u, 0u, 0u, 0u};
            if (r0 + row < mend && kc + col < K) xr[q] = *reinterpret_cast<const u32x4*>(x + (int64_t)(r0 + row) * ldx + kc + col);
        }
        if (cc == 0) {
            tr8 = zero8();
            if (tid < tvecs) {
                const int64_t e0 = (int64_t)r0 * r + tid * 8;
                if (e0 + 8 <= (int64_t)mend * r) {
                    tr8 = *reinterpret_cast<const bf16x8*>(t + e0);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (e0 + q < (int64_t)mend * r) tr8[q] = t[e0 + q];
                }
            }
        }
    };
    if (mstart < mend) fetch(mstart, 0);
    for (int r0 = mstart; r0 < mend; r0 += 64) {
        f32x4 ua = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
            if (cc >= nct) break;
            __syncthreads();  // previous tile fully consumed
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 256 + tid;
                *reinterpret_cast<u32x4*>(Xs + (c / 16) * WG_LD + (c % 16) * 8) = xr[q];
            }
            if (cc == 0 && tid < tvecs) *reinterpret_cast<bf16x8*>(Ts + tid * 8) = tr8;
            __syncthreads();
            // the next tile's loads fly under this tile's MFMAs
            if (cc + 1 < nct) fetch(r0, cc + 1);
            else if (r0 + 64 < mend) fetch(r0 + 64, 0);
            // this wave: output columns [wv*32, wv*32+32) of the tile, all 64 rows of the slab
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 b = zero8();
                if (j < r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] = Ts[(ks * 16 + 8 * (i >> 2) + 4 * h + (i & 3)) * r + j];
                }
                bf16x8 a = tr_frag(Xs, WG_LD, ks * 16, wv * 32, lane);
                acc[cc] = mfma32(a, b, acc[cc]);
            }
            if (with_u) {
                // wave wv takes slab rows 16 wv .. + 15 (the MFMA's n), the tile's 128 columns (k);
                // D[m = 4 (lane >> 4) + i][n = lane & 15]: a lane holds four consecutive rank entries of one token row
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 xb = *reinterpret_cast<const bf16x8*>(Xs + (wv * 16 + (lane & 15)) * WG_LD + ks * 32 + 8 * (lane >> 4));
                    ua = __builtin_amdgcn_mfma_f32_16x16x32_bf16(upf[cc][ks], xb, ua, 0, 0, 0);
                }
            }
        }
        if (with_u) {  // u_part[strip][row][rank]
            const int row = r0 + wv * 16 + (lane & 15), jq = 4 * (lane >> 4);
            if (row < mend && jq < r) *reinterpret_cast<f32x4*>(u_part + ((int64_t)blockIdx.x * M + row) * r + jq) = ua;
        }
    }
    if (j < r) {
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + cc * WG_COLS + wv * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (cc < nct && k < K) p.part[((int64_t)mb * K + k) * r + j] = acc[cc][i];
            }
    }
}

// stage 2: out_g(j, k) (+)= s · Σ_mb P[mb][k][j]   (bf16; straight into the gradient tensors, padded rank rows /
// off-diagonal blocks of a grouped problem dropped), both problems in one launch
__global__ __launch_bounds__(256) void lora_wgrad_reduce_kernel(WgradProblem p0, WgradProblem p1, int nmb, int r, float s,
                                                                int accumulate) {
    const WgradProblem& p = blockIdx.y == 0 ? p0 : p1;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ne = (int64_t)p.K * r;
    if (e >= ne) return;
    const int k = (int)(e / r), j = (int)(e % r);
    bf16_t* dst;
    if (p.kind == 0) {
        const int g = j / p.rsub;
        if (g >= p.groups) return;  // zero-padded rank rows
        dst = p.out[g] + (int64_t)(j - g * p.rsub) * p.K + k;
    } else {
        const int g = k / p.ksub, jj = j - g * p.rsub;
        if (jj < 0 || jj >= p.rsub) return;  // off the block diagonal / padded rank columns
        dst = p.out[g] + (int64_t)(k - g * p.ksub) * p.rsub + jj;
    }
    float v = 0.f;
    for (int mb = 0; mb < nmb; ++mb) v += p.part[(int64_t)mb * ne + e];
    v *= s;
    if (accumulate) v += (float)*dst;
    *dst = (bf16_t)v;
}

// u[M][r] (bf16) = s * sum over column tiles c (in order) of u_part[c][M][r]; one thread = four rank entries of a row
__global__ __launch_bounds__(256) void lora_u_reduce_kernel(const float* __restrict__ u_part, bf16_t* __restrict__ u, int64_t M,
                                                            int r, int ctiles, float s) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // float4 index
    const int64_t n4 = M * r / 4;
    if (e >= n4) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < ctiles; ++c) v += reinterpret_cast<const f32x4*>(u_part)[(int64_t)c * n4 + e];
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(v[i] * s);
    reinterpret_cast<bf16x4*>(u)[e] = o;
}

// ---------------------------------------------------------------------------------------------
// Adapter operand forms for the fused projection GEMM, all layers of a model in ONE launch (vft_lora_pack).
// The GEMM's rank extension wants lora_up as (N, rp) and — for the data gradient — lora_down transposed (K, rp);
// vft_lora_down wants lora_up transposed (rp, N); a fused q|k|v projection wants the three adapters concatenated
// (lora_down rows) / block-diagonal (lora_up), zero-padded to the extension rank rp.  Each is a tiny tensor, but built
// per layer and per step with tensor ops they are ~3 000 launches; here one descriptor per output tensor:
//   kind 0  dst (rp, L)      = [A_0; A_1; A_2; 0]                    A_g = src[g] (r, L)
//   kind 1  dst (L, rp)      = the transpose of kind 0
//   kind 2  dst (G L, rp)    = blockdiag(B_0, B_1, B_2) | 0           B_g = src[g] (L, r)
//   kind 3  dst (rp, G L)    = the transpose of kind 2
// ---------------------------------------------------------------------------------------------
struct PackDesc {
    const bf16_t* src[3];
    bf16_t* dst;
    int32_t kind, groups, r, rp;
    int64_t L;
};
static_assert(sizeof(PackDesc) == 56, "PackDesc is built by the host as 7 x int64");

__global__ __launch_bounds__(256) void lora_pack_kernel(const PackDesc* __restrict__ table) {
    const PackDesc d = table[blockIdx.y];
    const int64_t rows = (d.kind == 0 || d.kind == 3) ? d.rp : (d.kind == 1 ? d.L : d.groups * d.L);
    const int64_t cols = (d.kind == 0) ? d.L : (d.kind == 3 ? d.groups * d.L : d.rp);
    const int64_t n = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / cols, c = e - i * cols;
        // (j, l): rank index in the padded rank, position along L (kinds 0/1) or along G*L (kinds 2/3)
        const int64_t j = (d.kind == 0 || d.kind == 3) ? i : c;
        const int64_t l = (d.kind == 0 || d.kind == 3) ? c : i;
        bf16_t v = (bf16_t)0.f;
        if (d.kind <= 1) {
            const int g = (int)(j / d.r);
            if (g < d.groups) v = d.src[g][(j - (int64_t)g * d.r) * d.L + l];
        } else {
            const int g = (int)(l / d.L);
            const int64_t jj = j - (int64_t)g * d.r;
            if (jj >= 0 && jj < d.r) v = d.src[g][(l - (int64_t)g * d.L) * d.r + jj];
        }
        d.dst[e] = v;
    }
}

// every queued reduction in one launch: blockIdx.y = table entry, blockIdx.x over its K*r elements (entries with fewer
// elements than the grid's widest leave their surplus blocks at once)
__global__ __launch_bounds__(256) void lora_wgrad_reduce_many_kernel(const ReduceDesc* __restrict__ table) {
    const ReduceDesc d = table[blockIdx.y];
    const WgradProblem& p = d.p;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ne = (int64_t)p.K * d.r;
    if (e >= ne) return;
    const int k = (int)(e / d.r), j = (int)(e % d.r);
    bf16_t* dst;
    if (p.kind == 0) {
        const int g = j / p.rsub;
        if (g >= p.groups) return;
        dst = p.out[g] + (int64_t)(j - g * p.rsub) * p.K + k;
    } else {
        const int g = k / p.ksub, jj = j - g * p.rsub;
        if (jj < 0 || jj >= p.rsub) return;
        dst = p.out[g] + (int64_t)(k - g * p.ksub) * p.rsub + jj;
    }
    float v = 0.f;
    for (int mb = 0; mb < d.nmb; ++mb) v += p.part[(int64_t)mb * ne + e];
    v *= d.scale;
    if (d.accumulate) v += (float)*dst;
    *dst = (bf16_t)v;
}

int chk(bool c, const char* msg) {
    if (!c) vft_set_error("%s", msg);
    return c ? 0 : 1;
}

template <bool NJ>
int launch_rows(const void* x, void* y, const void* w1, const void* w2, void* p_out, void* w2t_out, const void* res,
                int64_t ldres, int M, int K, int N, int r, int64_t ldx, int64_t ldy, float s, hipStream_t st) {
    dim3 grid((M + 15) / 16), block(256);
#define VFT_ROWS(RC)                                                                                                  \
    hipLaunchKernelGGL((lora_rows_kernel<RC, NJ>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, (const bf16_t*)w1, \
                       (const bf16_t*)w2, (bf16_t*)p_out, (bf16_t*)w2t_out, (const bf16_t*)res, ldres, M, K, N, r, ldx, ldy, s)
    if (r <= 4)
        VFT_ROWS(4);
    else if (r <= 8)
        VFT_ROWS(8);
    else
        VFT_ROWS(16);
#undef VFT_ROWS
    return 0;
}

int check_rows(const char* name, const void* x, const void* y, int M, int K, int N, int r, int64_t ldx, int64_t ldy) {
    VFT_CHECK_ARG(M > 0 && K > 0 && N > 0 && r > 0 && r <= 16, "%s: need M,K,N>0 and 0<r<=16 (got M=%d K=%d N=%d r=%d)", name, M, K, N, r);
    VFT_CHECK_ARG(K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "%s: K, N and row strides must be multiples of 8", name);
    VFT_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "%s: activations must be 16-byte aligned", name);
    return 0;
}

}  // namespace

extern "C" int vft_lora_fwd(const void* x, void* y, const void* lora_down, const void* lora_up, void* t_out,
                            void* up_t_out, const void* residual, int M, int K, int N, int r, int64_t ldx, int64_t ldy,
                            int64_t ldres, float scale, void* stream) {
    if (chk(x && y && lora_down && lora_up, "vft_lora_fwd: null pointer")) return 1;
    if (check_rows("vft_lora_fwd", x, y, M, K, N, r, ldx, ldy)) return 1;
    VFT_CHECK_ARG(((uintptr_t)lora_down & 15) == 0 && ((uintptr_t)up_t_out & 15) == 0, "vft_lora_fwd: lora_down / up_t_out must be 16-byte aligned");
    VFT_CHECK_ARG(residual == nullptr || (ldres % 8 == 0 && ((uintptr_t)residual & 15) == 0), "vft_lora_fwd: bad residual pointer / stride");
    launch_rows<true>(x, y, lora_down, lora_up, t_out, up_t_out, residual, ldres, M, K, N, r, ldx, ldy, scale, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_lora_fwd");
    return 0;
}

// t[M][r] = x[M][K] . lora_down[r][K]^T only (the low-rank activations; the update itself is applied by the consumer:
// vft_geglu_lora_*, or the rank extension of vft_linear_fwd / vft_linear_nf4)
// A/B control (vft_lora_down_variant): 0 = automatic (32 rows per workgroup; 64 at rank > 8 and M >= 16384), 1 = the 16-row
// phase 1 of lora_rows_kernel, 2 = 64 rows, 3 = 32 rows
static int g_lora_down_variant = 0;
extern "C" int vft_lora_down_variant(int v) {
    g_lora_down_variant = v;
    return 0;
}

extern "C" int vft_lora_down(const void* x, const void* lora_down, void* t_out, int M, int K, int r, int64_t ldx,
                             float scale, void* stream) {
    if (chk(x && lora_down && t_out, "vft_lora_down: null pointer")) return 1;
    VFT_CHECK_ARG(M > 0 && K > 0 && r > 0 && r <= 16 && K % 8 == 0 && ldx % 8 == 0, "vft_lora_down: need M,K>0, 0<r<=16, K and ldx multiples of 8");
    VFT_CHECK_ARG((((uintptr_t)x | (uintptr_t)lora_down) & 15) == 0, "vft_lora_down: operands must be 16-byte aligned");
    if (g_lora_down_variant == 1)
        launch_rows<true>(x, nullptr, lora_down, nullptr, t_out, nullptr, nullptr, 0, M, K, 0, r, ldx, 0, scale, (hipStream_t)stream);
    // 64 rows per workgroup at rank > 8.  Round 2 measured them on inputs that sat in the Infinity Cache and found them better
    // only from M = 65 536 on; on COLD inputs — the in-step case, tools/exp/lora_down_cold.py cycles through 700 MB of them —
    // they are 10-20 % faster at AuraFlow's M = 26 160 too (K 3072: 48.7 -> 44.1 us, K 8192: 134 -> 112), and 25-45 % SLOWER
    // at rank 4 whatever M.  Tried in round 4 and dropped: a contiguous quarter of K per wave instead of interleaved k-steps
    // (10-20 % slower: four waves of a workgroup walking one row together keep the DRAM pages open)
    else if (g_lora_down_variant == 2 || (g_lora_down_variant == 0 && r > 8 && M >= 16384))
        hipLaunchKernelGGL(lora_down32_kernel<4>, dim3((M + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)lora_down, (bf16_t*)t_out, M, K, r, ldx, scale);
    else
        hipLaunchKernelGGL(lora_down32_kernel<2>, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)lora_down, (bf16_t*)t_out, M, K, r, ldx, scale);
    VFT_CHECK_LAUNCH("vft_lora_down");
    return 0;
}

extern "C" int vft_lora_bwd_dx(const void* dy, void* dx, const void* lora_up_t, const void* lora_down, void* u_out,
                               int M, int N, int K, int r, int64_t lddy, int64_t lddx, float scale, void* stream) {
    if (chk(dy && dx && lora_up_t && lora_down && u_out, "vft_lora_bwd_dx: null pointer")) return 1;
    if (check_rows("vft_lora_bwd_dx", dy, dx, M, N, K, r, lddy, lddx)) return 1;
    VFT_CHECK_ARG(((uintptr_t)lora_up_t & 15) == 0 && ((uintptr_t)lora_down & 15) == 0, "vft_lora_bwd_dx: weights must be 16-byte aligned");
    // phase 1 contracts over N with Bᵀ (r,N); phase 2 streams dx (M,K) with A (r,K) in JN layout
    launch_rows<false>(dy, dx, lora_up_t, lora_down, u_out, nullptr, nullptr, 0, M, N, K, r, lddy, lddx, scale, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_lora_bwd_dx");
    return 0;
}

extern "C" int vft_lora_wgrad_target(int wgs) {
    g_wgrad_target_wgs = wgs > 0 ? wgs : 0;
    return 0;
}

extern "C" size_t vft_lora_wgrad_workspace(int M, int K, int N, int r) {
    const int rpb = wgrad_rows_per_block(M, K, N);
    return (size_t)((M + rpb - 1) / rpb) * (size_t)(K + N) * (size_t)r * sizeof(float);
}

// which: 3 = both problems in one launch (the forms of rounds 1-4); 1 = only d(lora_down) = u^T x; 2 = only d(lora_up) = dy^T t,
// optionally with the u partial sums (up_t, u_part: see the kernel).  The workspace of a one-problem launch holds that
// problem's partial sums only.
static int wgrad_launch(const void* x, const void* u, const void* dy, const void* t, void* const* d_down, void* const* d_up,
                        int groups, int rsub, int accumulate, float* workspace, size_t ws_bytes, int M, int K, int N, int r,
                        int64_t ldx, int64_t lddy, float scale, void* stream, const char* name, void* defer_slot = nullptr,
                        int which = 3, const void* up_t = nullptr, float* u_part = nullptr) {
    if (chk(((which & 1) ? (x && u && d_down) : true) && ((which & 2) ? (dy && t && d_up) : true) && workspace, "vft_lora_wgrad2: null pointer")) return 1;
    VFT_CHECK_ARG(M > 0 && K > 0 && N > 0 && r > 0 && r <= 32, "%s: need M,K,N>0 and 0<r<=32", name);
    VFT_CHECK_ARG(K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0, "%s: K, N, strides must be multiples of 8", name);
    VFT_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy) & 15) == 0, "%s: activations must be 16-byte aligned", name);
    VFT_CHECK_ARG(groups >= 1 && groups <= 3 && rsub >= 1 && groups * rsub <= r && N % groups == 0,
                  "%s: need 1 <= groups <= 3, groups * rsub <= r and N %% groups == 0 (groups=%d rsub=%d r=%d N=%d)", name, groups, rsub, r, N);
    VFT_CHECK_ARG(which >= 1 && which <= 3, "%s: which = %d", name, which);
    VFT_CHECK_ARG((up_t == nullptr) == (u_part == nullptr) && (up_t == nullptr || (which == 2 && r % 4 == 0 && r <= 16 && (((uintptr_t)up_t | (uintptr_t)u_part) & 15) == 0)),
                  "%s: the u partial sums need the dy-only launch, a rank padded to 4 / 8 / 16 and 16-byte aligned up_t / u_part", name);
    const int rpb = wgrad_rows_per_block(M, (which & 1) ? K : 0, (which & 2) ? (u_part ? (N + UCT - 1) / UCT : N) : 0);
    const int nmb = (M + rpb - 1) / rpb;
    const size_t need = (size_t)nmb * ((which & 1 ? (size_t)K : 0) + (which & 2 ? (size_t)N : 0)) * (size_t)r * sizeof(float);
    VFT_CHECK_ARG(ws_bytes >= need, "%s: workspace too small (%zu < %zu bytes)", name, ws_bytes, need);
    WgradProblem pa{(const bf16_t*)x, (const bf16_t*)u, workspace, {nullptr, nullptr, nullptr}, K, ldx, 0, groups, rsub, 0};
    WgradProblem pb{(const bf16_t*)dy, (const bf16_t*)t, workspace + ((which & 1) ? (size_t)nmb * K * r : 0), {nullptr, nullptr, nullptr}, N, lddy,
                    1, groups, rsub, N / groups};
    for (int g = 0; g < groups; ++g) {
        if (chk(((which & 1) ? d_down[g] != nullptr : true) && ((which & 2) ? d_up[g] != nullptr : true), "vft_lora_wgrad2: null gradient pointer")) return 1;
        if (which & 1) pa.out[g] = (bf16_t*)d_down[g];
        if (which & 2) pb.out[g] = (bf16_t*)d_up[g];
    }
    const size_t lds = 64 * WG_LD * 2 + 64 * (size_t)r * 2;
    const int kmax = which == 3 ? (K > N ? K : N) : (which == 1 ? K : N);
    if (u_part != nullptr)
        hipLaunchKernelGGL(lora_wgrad_partial_kernel<UCT>, dim3((kmax + UCT * WG_COLS - 1) / (UCT * WG_COLS), nmb, 1), dim3(256), lds,
                           (hipStream_t)stream, pa, pb, M, r, rpb, (ReduceDesc*)defer_slot, nmb, scale, accumulate, 1, (const bf16_t*)up_t, u_part);
    else
        hipLaunchKernelGGL(lora_wgrad_partial_kernel<1>, dim3((kmax + WG_COLS - 1) / WG_COLS, nmb, which == 3 ? 2 : 1), dim3(256), lds,
                           (hipStream_t)stream, pa, pb, M, r, rpb, (ReduceDesc*)defer_slot, nmb, scale, accumulate, which == 2 ? 1 : 0,
                           (const bf16_t*)nullptr, (float*)nullptr);
    VFT_CHECK_LAUNCH("vft_lora_wgrad2(partial)");
    if (defer_slot != nullptr) return 0;  // the reduction is queued: vft_lora_wgrad_reduce_many runs it
    const int64_t ne = (int64_t)kmax * r;
    if (which == 3)
        hipLaunchKernelGGL(lora_wgrad_reduce_kernel, dim3((unsigned)((ne + 255) / 256), 2), dim3(256), 0,
                           (hipStream_t)stream, pa, pb, nmb, r, scale, accumulate);
    else  // one problem: it sits in slot 0 of the reduce kernel's pair
        hipLaunchKernelGGL(lora_wgrad_reduce_kernel, dim3((unsigned)((ne + 255) / 256), 1), dim3(256), 0,
                           (hipStream_t)stream, which == 1 ? pa : pb, pb, nmb, r, scale, accumulate);
    VFT_CHECK_LAUNCH("vft_lora_wgrad2(reduce)");
    return 0;
}

extern "C" int vft_lora_wgrad2(const void* x, const void* u, const void* dy, const void* t, void* d_down, void* d_up,
                               float* workspace, size_t ws_bytes, int M, int K, int N, int r, int64_t ldx, int64_t lddy,
                               float scale, void* stream) {
    void* dd[1] = {d_down};
    void* du[1] = {d_up};
    return wgrad_launch(x, u, dy, t, dd, du, 1, r, 0, workspace, ws_bytes, M, K, N, r, ldx, lddy, scale, stream, "vft_lora_wgrad2");
}

// Grouped form: `groups` adapters that share x (a fused q|k|v projection: u / t carry the groups' rank columns side by
// side, dy their output columns side by side), rank `rsub` each inside the padded rank r.  Gradients are written straight
// into the caller's tensors — d_down[g] (rsub, K), d_up[g] (N / groups, rsub), dense — and with accumulate != 0 added to
// what they hold (the trainer's flat gradient buffer: no autograd accumulation pass, no slicing of padded results).
extern "C" int vft_lora_wgrad2g(const void* x, const void* u, const void* dy, const void* t, void* const* d_down,
                                void* const* d_up, int groups, int rsub, int accumulate, float* workspace, size_t ws_bytes,
                                int M, int K, int N, int r, int64_t ldx, int64_t lddy, float scale, void* stream) {
    return wgrad_launch(x, u, dy, t, d_down, d_up, groups, rsub, accumulate, workspace, ws_bytes, M, K, N, r, ldx, lddy, scale,
                        stream, "vft_lora_wgrad2g");
}

// table: n descriptors of 7 x int64 in device memory: {src0, src1, src2, dst, kind | groups << 32, r | rp << 32, L}
// (see PackDesc above).  One launch rebuilds every derived adapter operand of a model; graph-capturable.
extern "C" int vft_lora_pack(const void* table, int n, void* stream) {
    if (chk(table != nullptr, "vft_lora_pack: null table")) return 1;
    VFT_CHECK_ARG(n > 0 && n <= 65535, "vft_lora_pack: need 0 < n <= 65535 descriptors (got %d)", n);
    hipLaunchKernelGGL(lora_pack_kernel, dim3(16, (unsigned)n), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)table);
    VFT_CHECK_LAUNCH("vft_lora_pack");
    return 0;
}

// Deferred form: only the partial sums are computed now; the two reductions (d_down, d_up) are written as entries
// `slot`, `slot + 1` of the caller's device table (vft_lora_wgrad_desc_bytes() bytes per entry) by the kernel itself, and
// ONE vft_lora_wgrad_reduce_many launch at the end of the backward pass runs every queued reduction — 560 + 560 launches
// per SDXL-LoRA step become 560 + 1.  The workspace must stay untouched until then.
extern "C" size_t vft_lora_wgrad_desc_bytes() { return sizeof(ReduceDesc); }

extern "C" int vft_lora_wgrad2g_defer(const void* x, const void* u, const void* dy, const void* t, void* const* d_down,
                                      void* const* d_up, int groups, int rsub, int accumulate, float* workspace,
                                      size_t ws_bytes, int M, int K, int N, int r, int64_t ldx, int64_t lddy, float scale,
                                      void* table, int slot, void* stream) {
    if (chk(table != nullptr && slot >= 0, "vft_lora_wgrad2g_defer: null table / negative slot")) return 1;
    return wgrad_launch(x, u, dy, t, d_down, d_up, groups, rsub, accumulate, workspace, ws_bytes, M, K, N, r, ldx, lddy, scale,
                        stream, "vft_lora_wgrad2g_defer", (char*)table + (size_t)slot * sizeof(ReduceDesc));
}

// ---- round 5: the backward's rank-sized work in two passes that read dy and x ONCE each ------------------------------------
// vft_lora_bwd_dy:  d(lora_up) partial sums (as vft_lora_wgrad2g[_defer], problem "dy^T t" only) AND u = scale_u * dy * lora_up
//                   (M, r) bf16 — the operand the data-gradient GEMM's rank extension needs — from the same pass over dy:
//                   per column tile partial sums in u_part (ctiles * M * r floats, ctiles = ceil(N / 128): vft_lora_u_part_floats),
//                   then an ordered sum.  Replaces vft_lora_down(dy, lora_up^T) + half of vft_lora_wgrad2g.
// vft_lora_bwd_x:   d(lora_down) partial sums (problem "u^T x" only) — the other half.
// table / slot: as vft_lora_wgrad2g_defer (one descriptor per call), or table == nullptr for an immediate reduction.
static int u_strips(int N) { return (N + UCT * WG_COLS - 1) / (UCT * WG_COLS); }  // partial-sum planes of u
extern "C" int64_t vft_lora_u_part_floats(int M, int N, int r) { return (int64_t)u_strips(N) * M * r; }
// bytes of partial sums of ONE of the two problems: cols = N, with_u = 1 for vft_lora_bwd_dy; cols = K, with_u = 0 for vft_lora_bwd_x
extern "C" size_t vft_lora_bwd_workspace(int M, int cols, int r, int with_u) {
    const int rpb = wgrad_rows_per_block(M, with_u ? (cols + UCT - 1) / UCT : cols, 0);
    return (size_t)((M + rpb - 1) / rpb) * (size_t)cols * (size_t)r * sizeof(float);
}

extern "C" int vft_lora_bwd_dy(const void* dy, const void* t, const void* up_t, void* u_out, float* u_part, void* const* d_up,
                               int groups, int rsub, int accumulate, float* workspace, size_t ws_bytes, int M, int N, int r,
                               int64_t lddy, float scale_u, float scale_w, void* table, int slot, void* stream) {
    if (chk(up_t && u_out && u_part, "vft_lora_bwd_dy: null pointer")) return 1;
    VFT_CHECK_ARG(((uintptr_t)u_out & 7) == 0, "vft_lora_bwd_dy: u_out must be 8-byte aligned");
    void* none[3] = {nullptr, nullptr, nullptr};
    if (wgrad_launch(nullptr, nullptr, dy, t, none, d_up, groups, rsub, accumulate, workspace, ws_bytes, M, /*K=*/8, N, r, 8, lddy, scale_w,
                     stream, "vft_lora_bwd_dy", table ? (char*)table + (size_t)slot * sizeof(ReduceDesc) : nullptr, 2, up_t, u_part))
        return 1;
    const int64_t n4 = (int64_t)M * r / 4;
    hipLaunchKernelGGL(lora_u_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u_part,
                       (bf16_t*)u_out, (int64_t)M, r, u_strips(N), scale_u);
    VFT_CHECK_LAUNCH("vft_lora_bwd_dy(u reduce)");
    return 0;
}

extern "C" int vft_lora_bwd_x(const void* x, const void* u, void* const* d_down, int groups, int rsub, int accumulate,
                              float* workspace, size_t ws_bytes, int M, int K, int r, int64_t ldx, float scale_w, void* table,
                              int slot, void* stream) {
    void* none[3] = {nullptr, nullptr, nullptr};
    return wgrad_launch(x, u, nullptr, nullptr, d_down, none, groups, rsub, accumulate, workspace, ws_bytes, M, K, /*N=*/8 * groups, r, ldx, 8,
                        scale_w, stream, "vft_lora_bwd_x", table ? (char*)table + (size_t)slot * sizeof(ReduceDesc) : nullptr, 1);
}

// n entries of the table, the widest of which has max_elements = K * r output elements
extern "C" int vft_lora_wgrad_reduce_many(const void* table, int n, int64_t max_elements, void* stream) {
    if (chk(table != nullptr, "vft_lora_wgrad_reduce_many: null table")) return 1;
    VFT_CHECK_ARG(n > 0 && n <= 65535 && max_elements > 0, "vft_lora_wgrad_reduce_many: need 0 < n <= 65535 entries (got %d)", n);
    hipLaunchKernelGGL(lora_wgrad_reduce_many_kernel, dim3((unsigned)((max_elements + 255) / 256), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, (const ReduceDesc*)table);
    VFT_CHECK_LAUNCH("vft_lora_wgrad_reduce_many");
    return 0;
}
