// Flash attention forward + backward for gfx950 (CDNA4), bf16 in/out, fp32 softmax/accumulate.
//
// Replaces the third-party kernels the reference dispatches to from
//   src/modules/attention.py:98-159 (scaled_dot_product_attention: F.sdpa / flash_attn_func / xformers)
//   src/modules/attention.py:31-95  (scaled_qkv_attention, AuraFlow entry)
// Non-causal, dropout 0, optional key-padding mask given as kv_len[b] (Lumina2's prefix bool mask,
// src/models/lumina2/denoiser.py:150-155), GQA by head-index division (Hq % Hkv == 0).
//
// Tiling (wave64, v_mfma_f32_32x32x16_bf16 everywhere):
//   * every product is arranged so the softmax row (the query) sits on the MFMA *lane* and the
//     reduction index sits in the accumulator *registers*:  S^T = K·Q^T  (keys in regs, q on lane).
//     Row max / row sum are then 16 in-lane ops + one lane^32 exchange, and the fp32 accumulator
//     tile is, after a pairwise bf16 pack, directly the B operand of the next product
//     (O^T = V^T·P^T, dQ^T = K^T·dS^T, dV^T = dO^T·P, dK^T = Q^T·dS) — P/dS never touch LDS.
//   * the A operand of those second products is a *column* read of a row-major LDS tile, done with
//     ds_read_b64_tr_b16 (hardware transpose), so K/V/Q/dO tiles are staged once, row-major.
//   * LDS rows are padded by 16 B: row stride (D+8)*2 B == 4*odd dwords, conflict-free for
//     ds_read_b128 fragment reads of 16 consecutive rows.
//   * global->LDS staging goes through registers and is issued one tile ahead of the compute
//     (issue loads early, write to LDS after the barrier).
//
// Backward is two kernels (no atomics, bitwise reproducible): dq (one workgroup per 128 queries, loops
// over key tiles; also produces delta = rowsum(dO*O)) and dkdv (one workgroup per 128 keys, loops over
// query tiles). Both recompute P from the saved log-sum-exp.
#include <type_traits>

#include "vft_common.h"
#include "wgrad8.h"

namespace {

// A/B build switch: 1 = the dq3 / dkdv3 backward kernels issue their LDS-DMA through the inline-asm groups too
#ifndef VFT_ATTN_BWD_ASM_DMA
#define VFT_ATTN_BWD_ASM_DMA 0
#endif

struct Str3 {
    int64_t b, h, s;  // element strides of batch, head, sequence; head_dim is contiguous
};

struct AttnParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;              // fwd: out; bwd: forward output (for delta)
    const bf16_t* dout;     // bwd only
    bf16_t* dq;
    bf16_t* dk;
    bf16_t* dv;
    float* lse;             // [B, Hq, Sq] natural-log LSE of scaled scores
    float* delta;           // [B, Hq, Sq] rowsum(dO*O)   (one-pass backward: its negative)
    float* nlse;            // one-pass backward: [B, Hq, Sq] -lse / scale (the S accumulators start from it)
    const int32_t* kv_len;  // [B] or nullptr
    float* split_ws;        // bwd, nsplit > 1: fp32 partial dK | dV, each [nsplit][B*Hkv][Sk][D]
    bf16_t* ds_t;           // bwd, head_dim >= 128 stored-dS form: dS^T [B*Hq][Sk][ds_ld] written by the dK/dV kernel, else null
    int64_t ds_ld;
    float* dq_ws;           // stored-dS form: fp32 partial tiles of the batched dQ GEMM
    int nsplit;
    int dbg;                // -DVFT_ATTN_PROBE builds only (tools/exp/attn_bwd1_probe.py): VFT_ATTN_DEBUG bits, WRONG results when non-zero
    int B, Hq, Hkv, Sq, Sk;
    float scale;
    Str3 sq, sk, sv, so, sdo, sdq, sdk, sdv;
};

// A/B control (vft_attn_variant): 0 = automatic, 1 = the 4-wave kernels for every head_dim
int g_attn_variant = 0;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Transposed A-operand fragment for a 32(out rows = tile columns) x 16(k = tile rows) block of a
// row-major [rows][ld] bf16 LDS tile.  The k order inside the step matches the packed-accumulator
// B operand: element j of lane half h is tile row  row0 + 8*(j>>2) + 4*h + (j&3).
// `tr_lane_off` is the lane-dependent part of the address (loop invariant); row0/col0 are compile-time
// constants at every call site and fold into the ds_read immediate offset.
__device__ __forceinline__ int tr_lane_off(int ld, int lane) {
    const int h = lane >> 5, i = lane & 15;
    return (4 * h + (i >> 2)) * ld + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
}
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* lane_base, int ld, int row0, int col0) {
    const bf16_t* a0 = lane_base + row0 * ld + col0;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * ld));
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// pack accumulator registers 8*s .. 8*s+7 to the bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s + j];
    return r;
}

// register-staged copy of a ROWS x D bf16 tile into a padded LDS tile, 256 threads.  All per-thread
// addressing is hoisted out of the tile loop: `init` computes this thread's global pointer / LDS slot /
// row once; per tile only a uniform offset is added (the address math used to be ~40 % of the loop's
// VALU instructions).
template <int D, int ROWS>
struct TileStage {
    static constexpr int CPR = D / 8;               // 16-byte chunks per row
    static constexpr int TOTAL = ROWS * CPR;
    static constexpr int NCH = (TOTAL + 255) / 256;  // chunks per thread (the last pass may be partial)
    static constexpr bool FULL = TOTAL % 256 == 0;
    static constexpr bool REG = FULL && (256 % CPR) == 0;  // chunk i of a thread = chunk 0 shifted by i*(256/CPR) rows
    static constexpr int NP = REG ? 1 : NCH;
    u32x4 r[NCH];
    const bf16_t* gp[NP];
    int lrow[NP];
    int loff[NP];
    int64_t rstride;

    __device__ __forceinline__ void init(const bf16_t* base, int64_t row_stride) {
        constexpr int LD = D + 8;
        rstride = row_stride;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = i * 256 + threadIdx.x;
            const int row = c / CPR, col = (c % CPR) * 8;
            gp[i] = base + (int64_t)row * row_stride + col;
            lrow[i] = row;
            loff[i] = row * LD + col;
        }
    }
    // rows [row0, row0 + ROWS) of the tensor; rows >= row_limit are zero-filled
    __device__ __forceinline__ void load(int row0, int row_limit) {
        const int64_t toff = (int64_t)row0 * rstride;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bf16_t* src;
            int row;
            if (REG) {
                src = gp[0] + toff + (int64_t)(i * (256 / CPR)) * rstride;
                row = lrow[0] + i * (256 / CPR);
            } else {
                src = gp[i % NP] + toff;
                row = lrow[i % NP];
            }
            if (!FULL && i * 256 + (int)threadIdx.x >= TOTAL) continue;  // partial last pass
            if (row0 + row < row_limit) {
                r[i] = *reinterpret_cast<const u32x4*>(src);
            } else {
                r[i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* lds) const {
        constexpr int LD = D + 8;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (!FULL && i * 256 + (int)threadIdx.x >= TOTAL) continue;
            const int off = REG ? loff[0] + i * (256 / CPR) * LD : loff[i % NP];
            *reinterpret_cast<u32x4*>(lds + off) = r[i];
        }
    }
};

__device__ __forceinline__ bf16x8 ldg_frag(const bf16_t* p, bool ok) {
    if (ok) return *reinterpret_cast<const bf16x8*>(p);
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
    return z;
}

// Make the compiler wait HERE for an ordinary (VGPR-destination) load: an asm statement that takes the value as an operand.
// Needed in front of every loop that keeps LDS-DMA in flight: hipcc cannot count its own loads past a DMA, so a fragment
// loaded before the loop and first used inside it costs an `s_waitcnt vmcnt(0)` INSIDE the loop — which drains the DMA
// ring on every iteration (seen in the ISA of the dK/dV kernel: a four-buffer ring that never had more than one tile in
// flight).  After the pin no ordinary load is pending and the loop's only vmcnt waits are the counted ones written here.
template <class T>
__device__ __forceinline__ void pin_loaded(T& x) {
    asm volatile("" : "+v"(x));
}

__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    v[2] = (bf16_t)c;
    v[3] = (bf16_t)d;
    *reinterpret_cast<bf16x4*>(p) = v;
}

// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (block id % 8), each with
// its own L2.  All tiles of one (batch, head) stream the same K/V (or Q/dO) panels, so they are given
// block ids that are congruent mod 8: heads are taken in groups of 8 (one per XCD) and a group's tiles
// are interleaved.  Placement only affects speed; the mapping is a bijection onto (tile, head) pairs
// plus idle tail blocks when the head count is not a multiple of 8.
__device__ __forceinline__ bool xcd_map(int ntiles, int nheads, int& tile, int& head) {
    const int L = blockIdx.x;
    const int per_group = 8 * ntiles;
    const int g = L / per_group, rem = L % per_group;
    head = g * 8 + (rem & 7);
    tile = rem >> 3;
    return head < nheads;
}
__host__ inline unsigned xcd_grid(int ntiles, int nheads) { return (unsigned)(((nheads + 7) / 8) * 8 * ntiles); }

// =============================================================================================
// forward: one workgroup = 128 queries (4 waves x 32), key tiles of 64
// =============================================================================================
// (D <= 96: ask for 2 waves/SIMD => a 256-register budget, which makes hipcc emit the VGPR-destination
// form of the MFMAs; with the default 512-register budget it parks S/P in AGPRs and pays ~130
// v_accvgpr_read/write per key tile to get them to the VALU and back.)
template <int D>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_fwd_kernel(AttnParams p) {
    constexpr int LD = D + 8, KS = D / 16, DT = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + 64 * LD;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    int qtile, bh;
    if (!xcd_map((p.Sq + 127) / 128, p.B * p.Hq, qtile, bh)) return;  // block-uniform
    const int b = bh / p.Hq, hq = bh % p.Hq, q0 = qtile * 128;
    const int hkv = hq / (p.Hq / p.Hkv);
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Qb = p.q + b * p.sq.b + hq * p.sq.h;
    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* Ob = p.o + b * p.so.b + hq * p.so.h;

    const int qrow = q0 + w * 32 + r;
    const bool qok = qrow < p.Sq;
    bf16x8 Qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Qf[ks] = ldg_frag(Qb + (int64_t)qrow * p.sq.s + ks * 16 + 8 * h, qok);

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m = -1e30f, lsum = 0.f;
    const float sc = p.scale * kLog2e;

    const int ntiles = (kvl + 63) / 64;
    TileStage<D, 64> kst, vst;
    kst.init(Kb, p.sk.s);
    vst.init(Vb, p.sv.s);
    // K/V tiles are double-buffered in LDS: tile t+1 is written into the other buffer in the middle of
    // tile t's compute (its global loads were issued one iteration earlier), so there is a single barrier
    // per tile and neither HBM latency nor the LDS stores sit on the MFMA critical path.
    constexpr int BUF = 2 * 64 * LD;
    if (ntiles > 0) {
        kst.load(0, kvl);
        vst.load(0, kvl);
        kst.store(Ks);
        vst.store(Vs);
    }
    __syncthreads();
    if (ntiles > 1) {
        kst.load(64, kvl);
        vst.load(64, kvl);
    }
    const int kfrag_off = r * LD + 8 * h;        // row-fragment offset of this lane
    const int vtr_off = tr_lane_off(LD, lane);   // transposed-fragment offset of this lane

    for (int t = 0; t < ntiles; ++t) {
        const bf16_t* Kc = Ks + (t & 1) * BUF;
        const bf16_t* kfrag = Kc + kfrag_off;
        const bf16_t* vtr = Kc + 64 * LD + vtr_off;

        // S^T[key][q] = K Q^T
        f32x16 S[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[kt][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 a = *reinterpret_cast<const bf16x8*>(kfrag + kt * 32 * LD + ks * 16);
                S[kt] = mfma32(a, Qf[ks], S[kt]);
            }
        }
        const int n0 = t * 64;
        if (n0 + 64 > kvl) {  // block-uniform: only the last key tile can be ragged
            // the empty asm keeps this a real branch (otherwise it is if-converted into ~100 selects per tile)
            asm volatile("; ragged tail tile" ::: "memory");
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = n0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= kvl) S[kt][i] = -INFINITY;
                }
        }
        float tmax = S[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, S[kt][i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax * sc);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(S[kt][i] * sc - m_new);
                psum += pv;
                S[kt][i] = pv;
            }
        lsum = lsum * alpha + psum;
        m = m_new;
        // exact skip: once the running maxima have settled no lane needs a rescale (alpha == 1 everywhere)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
        }

        bf16x8 Pf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) Pf[kk] = pack8(S[kk >> 1], kk & 1);

        if (t + 1 < ntiles) {  // stage tile t+1 into the other buffer; then start fetching tile t+2
            bf16_t* Kn = Ks + ((t + 1) & 1) * BUF;
            kst.store(Kn);
            vst.store(Kn + 64 * LD);
            if (t + 2 < ntiles) {
                kst.load((t + 2) * 64, kvl);
                vst.load((t + 2) * 64, kvl);
            }
        }

        // O^T[d][q] += V^T P^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a = tr_frag(vtr, LD, kk * 16, dt * 32);
                O[dt] = mfma32(a, Pf[kk], O[dt]);
            }
        __syncthreads();
    }

    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (qok) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16_t* dst = Ob + (int64_t)qrow * p.so.s + dt * 32 + 8 * g + 4 * h;
                store4(dst, O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv, O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv);
            }
        if (h == 0 && p.lse) {
            p.lse[((int64_t)b * p.Hq + hq) * p.Sq + qrow] = ltot > 0.f ? (m + __log2f(ltot)) * kLn2 : -INFINITY;
        }
    }
}


// =============================================================================================
// forward, head_dim >= 128: one workgroup = 128 queries x 8 waves.  Wave w serves queries 32 (w & 3) .. +31 and
// the key half (w >> 2) of every 64-key tile, so the waves w and w + 4 — which share a SIMD — hold the same queries
// against different keys: each keeps its own running maximum / sum / output accumulator (O: D/2 registers, Q
// fragments: D/4 — two waves fit a SIMD where the 4-wave kernel above needs the whole register file for one) and the
// pair is merged once, through LDS, at the end (the split-KV identity  O = (a0 O0 + a1 O1) / (a0 l0 + a1 l1),
// a_g = 2^(m_g - max m)).  K/V tiles arrive by LDS-DMA (buffer_load ... lds: no staging registers, masked key rows
// read zeros through the descriptor's bounds check), double-buffered, one barrier per tile.  The LDS image is
// unpadded (a DMA writes lane-linear) with the 16-byte chunk c of row r stored at chunk
//     (c & ~15) | ((c & 15) ^ f(r)),   f(r) = ((r & 3) << 2) | ((r >> 2) & 3)
// which serves both the row reads of K (ds_read_b128, 32 rows x one chunk) and the transposed reads of V
// (ds_read_b64_tr_b16) without bank conflicts; the swizzle sits on the DMA's per-lane SOURCE chunk.
// =============================================================================================
// LDS-DMA (buffer_load ... offen lds) issued through INLINE ASM, on purpose.  With the builtin the compiler knows an LDS write
// is pending on the VM counter, cannot tell what it aliases, and puts `s_waitcnt vmcnt(0)` in front of the first
// ds_read_b64_tr_b16 that follows (seen in the ISA of every kernel here that mixes the two: the dK/dV kernel's ring of four
// buffers drained on every tile, one L2 round trip per tile).  Hidden in asm, the DMA is invisible to that bookkeeping: the
// kernels' own counted `s_waitcnt vmcnt(N)` + barrier (which they had all along) are then the only waits.
//   * One statement issues a GROUP of N pieces (N x 1 KiB: consecutive LDS destinations, one descriptor): M0 (the LDS
//     destination, compiler-reserved) is saved and restored once per group; `s_nop 4` covers a descriptor / base SGPR written
//     just before (SALU or v_readfirstlane -> VMEM read: 5 wait states), `s_nop 0` each M0 write -> LDS-DMA hazard.
//   * No "memory" clobber: a DMA only has to stay behind the wait + barrier that freed its destination, and those statements
//     are volatile asm WITH the clobber (volatile statements keep their order; the compiler's own LDS accesses cannot cross
//     the clobbering one).  Without the clobber the compiler may still schedule its LDS reads and MFMAs around the group.
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lds_void_t*)p; }
#define VFT_DMA_ONE(k, KB_) "s_add_u32 m0, %1, " #KB_ "\n\ts_nop 0\n\tbuffer_load_dwordx4 %" #k ", %2, 0 offen lds\n\t"
template <int N>
__device__ __forceinline__ void dma16_group(__amdgpu_buffer_rsrc_t r, char* l, const int (&voff)[N]) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8, "pieces per group");
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(l));
    unsigned keep;
    if constexpr (N == 1) {
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t" VFT_DMA_ONE(3, 0) "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "s"(r), "v"(voff[0]) : "scc");
    } else if constexpr (N == 2) {
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t" VFT_DMA_ONE(3, 0) VFT_DMA_ONE(4, 1024) "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "s"(r), "v"(voff[0]), "v"(voff[1]) : "scc");
    } else if constexpr (N == 4) {
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t" VFT_DMA_ONE(3, 0) VFT_DMA_ONE(4, 1024) VFT_DMA_ONE(5, 2048) VFT_DMA_ONE(6, 3072) "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "s"(r), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]) : "scc");
    } else {
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t" VFT_DMA_ONE(3, 0) VFT_DMA_ONE(4, 1024) VFT_DMA_ONE(5, 2048) VFT_DMA_ONE(6, 3072)
                     VFT_DMA_ONE(7, 4096) VFT_DMA_ONE(8, 5120) VFT_DMA_ONE(9, 6144) VFT_DMA_ONE(10, 7168) "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(dst), "s"(r), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7])
                     : "scc");
    }
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* l, int voff) {
    const int v[1] = {voff};
    dma16_group<1>(r, l, v);
}
// (256-byte and longer rows: 16 chunks per bank window, f as above; 128-byte rows (head_dim 64): two rows share a bank window
// and a row has 8 chunks: f(r) = (((r >> 1) & 1) << 2) | ((r >> 2) & 3) — the same two access patterns, conflict-free)
template <int RB>
__device__ __forceinline__ int swz_f(int row) {
    static_assert(RB == 128 || RB % 256 == 0, "row size");
    return RB == 128 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3));
}

// rows x (D*2)-byte tile by LDS-DMA, 512 threads: wave-instruction i covers 1024 bytes = 1024/(2D) rows.  Nothing is
// kept per instruction: the lane's row and swizzled source chunk are recomputed at issue (a few VALU operations per
// 1 KiB moved; the register file is what these kernels are short of).
template <int D, int ROWS, int NW = 8, bool ASM = true>
struct DmaTile {
    static constexpr int RB = D * 2;                  // bytes per row
    static constexpr int NI = ROWS * RB / 1024 / NW;  // instructions per wave per tile
    static constexpr int OOBV = (int)0x80000000u;
    // rows [row0, row0 + ROWS) of the tensor behind `rs` (row stride in elements); rows >= row_limit read zeros
    static __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, char* lds_tile, int row0, int row_limit, int row_stride_elems) {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        int voffs[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int byte = (wv * NI + q) * 1024 + lane * 16;
            const int r = byte / RB, slot = (byte % RB) >> 4;
            const int c = (slot & ~15) | ((slot & 15) ^ swz_f<RB>(r));
            const int voff = (row0 + r) * row_stride_elems * 2 + c * 16;
            voffs[q] = row0 + r < row_limit ? voff : OOBV;
        }
        if constexpr (ASM) {
            dma16_group<NI>(rs, lds_tile + wv * NI * 1024, voffs);
        } else {  // the builtin form: the compiler sees the pending LDS writes (see above).  Kept for the head_dim 64 dq / dkdv
                  // pair (no difference measured there).  At head_dim >= 128 the asm groups first measured SLOWER (AuraFlow
                  // backward 6.25 ms against 5.97) while every MFMA still waited for its own LDS read; with the explicit
                  // fragment pipeline of the dK/dV kernel in place they win (5.71 against 5.88), so they are the default there
#pragma unroll
            for (int q = 0; q < NI; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds_tile + (wv * NI + q) * 1024), 16, voffs[q], 0, 0, 0);
        }
    }
};

// The same tile image with NOTHING recomputed per tile on the vector ALU but one xor + one add per piece: the descriptor's
// base is moved to the tile's first row (scalar arithmetic) and its extent ends with the tensor, so rows past the end fail
// the hardware range check by themselves; the lane keeps its row offset (lane_row() * row stride) and its swizzled chunk of
// piece 0 — piece q's row is RPI * q further down and its swizzle differs from piece 0's by a compile-time constant
// (no carries between the lane / wave bits and q's bits of the row number, checked below).  For the one-wave-per-SIMD
// kernels (head_dim >= 128) where every vector instruction of the staging code is exposed.
constexpr int swz_rows_const(int RB, int row) { return RB == 128 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3)); }
template <int D, int ROWS, int NW>
struct DmaTileRel {
    static constexpr int RB = D * 2, NI = ROWS * RB / 1024 / NW, RPI = 1024 / RB;
    static_assert(RB % 256 == 0 && NI >= 1 && NI * RPI <= 8 && (NI * RPI & (NI * RPI - 1)) == 0, "piece rows must stay below the wave's row bits");
    static __device__ __forceinline__ int lane_row() {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        return wv * NI * RPI + (lane * 16) / RB;
    }
    static __device__ __forceinline__ int lane_chunk_bytes() {
        const int lane = threadIdx.x & 63, slot = ((lane * 16) % RB) >> 4;
        return ((slot & ~15) | ((slot & 15) ^ swz_f<RB>(lane_row()))) << 4;
    }
    static __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs_tile, char* lds_tile, int lane_row_bytes, int lane_chunk, int stride_bytes) {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        int voffs[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) voffs[q] = lane_row_bytes + q * RPI * stride_bytes + (lane_chunk ^ (swz_rows_const(RB, q * RPI) << 4));
        dma16_group<NI>(rs_tile, lds_tile + wv * NI * 1024, voffs);
    }
};

template <int D>
__global__ __launch_bounds__(512, 2) void attn_fwd2_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, RB = D * 2, TILE = 64 * RB;  // bytes of one K (or V) tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), qg = w & 3, g = w >> 2;
    int qtile, bh;
    if (!xcd_map((p.Sq + 127) / 128, p.B * p.Hq, qtile, bh)) return;  // block-uniform
    const int b = bh / p.Hq, hq = bh % p.Hq, q0 = qtile * 128;
    const int hkv = hq / (p.Hq / p.Hkv);
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Qb = p.q + b * p.sq.b + hq * p.sq.h;
    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* Ob = p.o + b * p.so.b + hq * p.so.h;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((int64_t)(p.Sk - 1) * p.sk.s + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((int64_t)(p.Sk - 1) * p.sv.s + D) * 2), 0x00020000);

    const int qrow = q0 + qg * 32 + r;
    const bool qok = qrow < p.Sq;
    bf16x8 Qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Qf[ks] = ldg_frag(Qb + (int64_t)qrow * p.sq.s + ks * 16 + 8 * h, qok);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pin_loaded(Qf[ks]);

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m = -1e30f, lsum = 0.f;
    const float sc = p.scale * kLog2e;

    const int ntiles = (kvl + 63) / 64;
    typedef DmaTile<D, 64> Stage;
    const int ks_s = (int)p.sk.s, vs_s = (int)p.sv.s;
    if (ntiles > 0) {
        Stage::issue(rk, smem, 0, kvl, ks_s);
        Stage::issue(rv, smem + TILE, 0, kvl, vs_s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment addresses of this lane (bytes inside a tile).  K rows: row 32 g + r, chunk 2 ks + h;
    // V transposed: rows 32 g + 16 kk + 4 h + (i >> 2) (+8), chunk 4 dt + 2 ((lane >> 4) & 1) + ((i & 3) >> 1)
    const int krow = 32 * g + r;
    const int kbase = krow * RB + ((h ^ swz_f<RB>(krow)) << 4);  // ^ ((ks & 7) << 5), + (ks >> 3) * 256
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int vbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 32 * g + 4 * h + (i16 >> 2) + 8 * hi;
        vbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;  // ^ ((dt & 3) << 6), + (dt >> 2) * 256, + kk * 16 * RB
    }

    for (int t = 0; t < ntiles; ++t) {
        const char* Kc = smem + (t & 1) * 2 * TILE;
        const char* Vc = Kc + TILE;
        if (t + 1 < ntiles) {  // the other buffer was last read before the barrier that ended tile t-1
            char* Kn = smem + ((t + 1) & 1) * 2 * TILE;
            Stage::issue(rk, Kn, (t + 1) * 64, kvl, ks_s);
            Stage::issue(rv, Kn + TILE, (t + 1) * 64, kvl, vs_s);
        }
        // S^T[key][q] = K Q^T for this wave's 32 keys
        f32x16 S;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kc + ((kbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256));
            S = mfma32(a, Qf[ks], S);
        }
        const int n0 = t * 64 + 32 * g;
        if (t * 64 + 64 > kvl) {  // block-uniform: only the last key tile can be ragged
            asm volatile("; ragged tail tile" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = n0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= kvl) S[i] = -INFINITY;
            }
        }
        float tmax = S[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, S[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // Lazy rescale: the reference exponent m only follows the running maximum when it has grown by more than 2^8
        // (P then stays below 256: nothing for bf16 / fp32 accumulation), so the D/2-register rescale of O — as much
        // VALU work as the tile's softmax — runs in the first tiles only instead of in most of them (with 32 queries per
        // wave SOME row sets a new maximum in nearly every tile).  m, l and O stay mutually consistent, so the result is
        // the same softmax; the pair merge and the LSE use m as it is.
        const float ts = tmax * sc;
        const float m_new = ts > m + 8.f ? ts : m;
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pv = __builtin_amdgcn_exp2f(S[i] * sc - m_new);
            psum += pv;
            S[i] = pv;
        }
        lsum = lsum * alpha + psum;
        m = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // exact skip once the running maxima have settled
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
        }
        bf16x8 Pf[2];
        Pf[0] = pack8(S, 0);
        Pf[1] = pack8(S, 1);
        // O^T[d][q] += V^T P^T over this wave's 32 keys
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const char* a0 = Vc + kk * 16 * RB + (dt >> 2) * 256;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (vbase[0] ^ ((dt & 3) << 6))));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (vbase[1] ^ ((dt & 3) << 6))));
                s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                O[dt] = mfma32(__builtin_bit_cast(bf16x8, both), Pf[kk], O[dt]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (issued a whole tile of compute ago)
        __builtin_amdgcn_s_barrier();
    }

    // ---- merge the wave pair: wave g finalises d-tiles [g DT/2, (g+1) DT/2); it receives the partner's accumulators for
    //      those tiles (and the partner's m, l) through the K/V buffers, which are idle now ------------------------------
    constexpr int HT = DT / 2;
    float4* X = reinterpret_cast<float4*>(smem);           // [8 waves][HT*4 float4][64 lanes]
    float2* ML = reinterpret_cast<float2*>(smem + 8 * HT * 4 * 64 * 16);  // [8 waves][64 lanes]
    const float ltot_own = lsum + __shfl_xor(lsum, 32, 64);
    // (register arrays must be indexed by compile-time constants: the two roles are separate code paths)
    auto give = [&](auto GIVE0) {  // write the accumulators of the partner's d-tiles
        constexpr int base = decltype(GIVE0)::value * HT;
#pragma unroll
        for (int j = 0; j < HT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                X[(w * HT * 4 + j * 4 + e) * 64 + lane] =
                    make_float4(O[base + j][4 * e], O[base + j][4 * e + 1], O[base + j][4 * e + 2], O[base + j][4 * e + 3]);
    };
    if (g == 0) give(std::integral_constant<int, 1>{}); else give(std::integral_constant<int, 0>{});
    ML[w * 64 + lane] = make_float2(m, ltot_own);
    __syncthreads();
    const int pw = w ^ 4;
    const float2 pml = ML[pw * 64 + lane];
    const float mm = fmaxf(m, pml.x);
    const float a_own = __builtin_amdgcn_exp2f(m - mm), a_par = __builtin_amdgcn_exp2f(pml.x - mm);
    const float ltot = ltot_own * a_own + pml.y * a_par;
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    auto take = [&](auto MINE) {  // combine, normalise and store this wave's d-tiles
        constexpr int base = decltype(MINE)::value * HT;
#pragma unroll
        for (int j = 0; j < HT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 po = X[(pw * HT * 4 + j * 4 + e) * 64 + lane];
                bf16_t* dst = Ob + (int64_t)qrow * p.so.s + (base + j) * 32 + 8 * e + 4 * h;
                store4(dst, (O[base + j][4 * e] * a_own + po.x * a_par) * inv, (O[base + j][4 * e + 1] * a_own + po.y * a_par) * inv,
                       (O[base + j][4 * e + 2] * a_own + po.z * a_par) * inv, (O[base + j][4 * e + 3] * a_own + po.w * a_par) * inv);
            }
    };
    if (qok) {
        if (g == 0) take(std::integral_constant<int, 0>{}); else take(std::integral_constant<int, 1>{});
        if (g == 0 && h == 0 && p.lse) {
            p.lse[((int64_t)b * p.Hq + hq) * p.Sq + qrow] = ltot > 0.f ? (mm + __log2f(ltot)) * kLn2 : -INFINITY;
        }
    }
}

// =============================================================================================
// forward, head_dim 64 / 96, round 4: the 4-wave tiling of attn_fwd_kernel (128 queries per workgroup, 32 per wave, every
// wave against all 64 keys of a tile, S^T = K Q^T with the softmax row on the lane, P from the accumulators straight into
// O^T = V^T P^T) on the operand path of the newer kernels: K / V tiles arrive by asm LDS-DMA into the unpadded swizzled
// image (head_dim 96: rows padded to 256 bytes, pad chunks requested out of range), double-buffered, one barrier per tile.
// The round-1 kernel staged K / V through registers: its loop carried ~100 vector instructions of load addressing, range
// checks and LDS stores per 64-key tile beside the ~160 of the softmax (ISA count, round 4) — on a kernel that is bound by
// vector issue (16 MFMAs = 512 cycles against ~1 450 of VALU per tile and wave), and its padded image cost 24 % of the LDS
// cycles in bank conflicts on the transposed V reads.  Lazy rescale of O as in attn_fwd2_kernel.
// =============================================================================================
template <int D, int ROWS, int NW>
struct DmaTilePad;
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd3_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, RB = D == 96 ? 256 : D * 2, TILE = 64 * RB;  // bytes of one K (or V) tile
    typedef typename std::conditional<D == 96, DmaTilePad<D, 64, 4>, DmaTile<(D == 96 ? 128 : D), 64, 4>>::type Stage;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qtile, bh;
    if (!xcd_map((p.Sq + 127) / 128, p.B * p.Hq, qtile, bh)) return;  // block-uniform
    const int b = bh / p.Hq, hq = bh % p.Hq, q0 = qtile * 128;
    const int hkv = hq / (p.Hq / p.Hkv);
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Qb = p.q + b * p.sq.b + hq * p.sq.h;
    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* Ob = p.o + b * p.so.b + hq * p.so.h;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((int64_t)(p.Sk - 1) * p.sk.s + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((int64_t)(p.Sk - 1) * p.sv.s + D) * 2), 0x00020000);

    const int qrow = q0 + w * 32 + r;
    const bool qok = qrow < p.Sq;
    bf16x8 Qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Qf[ks] = ldg_frag(Qb + (int64_t)qrow * p.sq.s + ks * 16 + 8 * h, qok);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pin_loaded(Qf[ks]);

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m = -1e30f, lsum = 0.f;
    const float sc = p.scale * kLog2e;

    const int ntiles = (kvl + 63) / 64;
    const int ks_s = (int)p.sk.s, vs_s = (int)p.sv.s;
    if (ntiles > 0) {
        Stage::issue(rk, smem, 0, kvl, ks_s);
        Stage::issue(rv, smem + TILE, 0, kvl, vs_s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment addresses of this lane (bytes inside a tile): K rows r (+32 kt), chunk 2 ks + h; V transposed: rows
    // 16 kk + 4 h + (i >> 2) (+8), chunk 4 dt + 2 ((lane >> 4) & 1) + ((i & 3) >> 1) — the image of the other kernels
    const int rbase = r * RB + ((h ^ swz_f<RB>(r)) << 4);
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int tbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 4 * h + (i16 >> 2) + 8 * hi;
        tbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;
    }

    for (int t = 0; t < ntiles; ++t) {
        const char* Kc = smem + (t & 1) * 2 * TILE;
        const char* Vc = Kc + TILE;
        if (t + 1 < ntiles) {  // the other buffer was last read before the barrier that ended tile t-1
            char* Kn = smem + ((t + 1) & 1) * 2 * TILE;
            Stage::issue(rk, Kn, (t + 1) * 64, kvl, ks_s);
            Stage::issue(rv, Kn + TILE, (t + 1) * 64, kvl, vs_s);
        }
        // S^T[key][q] = K Q^T (rows r + 32 kt of the tile: 32 kt RB is a multiple of the swizzle period)
        f32x16 S[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[kt][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kc + kt * 32 * RB + ((rbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256));
                S[kt] = mfma32(a, Qf[ks], S[kt]);
            }
        }
        const int n0 = t * 64;
        if (n0 + 64 > kvl) {  // block-uniform: only the last key tile can be ragged
            asm volatile("; ragged tail tile" ::: "memory");
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = n0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= kvl) S[kt][i] = -INFINITY;
                }
        }
        float tmax = S[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, S[kt][i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // lazy rescale (see attn_fwd2_kernel): m follows the running maximum only when it has grown by more than 2^8
        const float ts = tmax * sc;
        const float m_new = ts > m + 8.f ? ts : m;
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(S[kt][i] * sc - m_new);
                psum += pv;
                S[kt][i] = pv;
            }
        lsum = lsum * alpha + psum;
        m = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // exact skip once the running maxima have settled
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
        }
        bf16x8 Pf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) Pf[kk] = pack8(S[kk >> 1], kk & 1);
        // O^T[d][q] += V^T P^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const char* a0 = Vc + kk * 16 * RB + (dt >> 2) * 256;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (tbase[0] ^ ((dt & 3) << 6))));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (tbase[1] ^ ((dt & 3) << 6))));
                s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                O[dt] = mfma32(__builtin_bit_cast(bf16x8, both), Pf[kk], O[dt]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (issued a whole tile of compute ago)
        __builtin_amdgcn_s_barrier();
    }

    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (qok) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16_t* dst = Ob + (int64_t)qrow * p.so.s + dt * 32 + 8 * g + 4 * h;
                store4(dst, O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv, O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv);
            }
        if (h == 0 && p.lse) {
            p.lse[((int64_t)b * p.Hq + hq) * p.Sq + qrow] = ltot > 0.f ? (m + __log2f(ltot)) * kLn2 : -INFINITY;
        }
    }
}

// Staggered form of attn_fwd2_kernel (same tiling, LDS image and pair merge): measured on MI355X it is 14-15 % faster at
// head_dim 128 (574 -> 655-663 TF/s, B16 H16 S2048) and 13 % SLOWER at head_dim 256, where the second copy of the loop body
// costs 24 spilled registers (728 -> 617 TF/s) — so it serves head_dim 128 only.
template <int D>
__global__ __launch_bounds__(512, 2) void attn_fwd2s_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, RB = D * 2, TILE = 64 * RB;  // bytes of one K (or V) tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), qg = w & 3, g = w >> 2;
    int qtile, bh;
    if (!xcd_map((p.Sq + 127) / 128, p.B * p.Hq, qtile, bh)) return;  // block-uniform
    const int b = bh / p.Hq, hq = bh % p.Hq, q0 = qtile * 128;
    const int hkv = hq / (p.Hq / p.Hkv);
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Qb = p.q + b * p.sq.b + hq * p.sq.h;
    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* Ob = p.o + b * p.so.b + hq * p.so.h;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((int64_t)(p.Sk - 1) * p.sk.s + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((int64_t)(p.Sk - 1) * p.sv.s + D) * 2), 0x00020000);

    const int qrow = q0 + qg * 32 + r;
    const bool qok = qrow < p.Sq;
    bf16x8 Qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Qf[ks] = ldg_frag(Qb + (int64_t)qrow * p.sq.s + ks * 16 + 8 * h, qok);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pin_loaded(Qf[ks]);

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    float m = -1e30f, lsum = 0.f;
    const float sc = p.scale * kLog2e;

    const int ntiles = (kvl + 63) / 64;
    typedef DmaTile<D, 64> Stage;
    const int ks_s = (int)p.sk.s, vs_s = (int)p.sv.s;
    if (ntiles > 0) {
        Stage::issue(rk, smem, 0, kvl, ks_s);
        Stage::issue(rv, smem + TILE, 0, kvl, vs_s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment addresses of this lane (bytes inside a tile).  K rows: row 32 g + r, chunk 2 ks + h;
    // V transposed: rows 32 g + 16 kk + 4 h + (i >> 2) (+8), chunk 4 dt + 2 ((lane >> 4) & 1) + ((i & 3) >> 1)
    const int krow = 32 * g + r;
    const int kbase = krow * RB + ((h ^ swz_f<RB>(krow)) << 4);  // ^ ((ks & 7) << 5), + (ks >> 3) * 256
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int vbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 32 * g + 4 * h + (i16 >> 2) + 8 * hi;
        vbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;  // ^ ((dt & 3) << 6), + (dt >> 2) * 256, + kk * 16 * RB
    }

    // The two wave groups (g = 0: waves 0-3, g = 1: waves 4-7; a SIMD holds one wave of each) run HALF A TILE APART:
    //     phase A(t):  g0  S = K(t) Q^T            | g1  softmax + O += V(t-1)^T P^T
    //     phase B(t):  g0  softmax + O += V(t)^T P | g1  S = K(t) Q^T
    // so in every phase one wave of a SIMD starts on the matrix pipe while its partner is in the softmax's VALU work,
    // instead of both waves doing the same thing at the same time (two waves in lockstep share the pipe they both want
    // and leave the other one idle).  K(t+1) is issued at the start of phase A(t) (its buffer's last reader, g1's S of
    // tile t-1, ended with phase B(t-1)), V(t+1) at the start of phase B(t) (last reader: g1's PV of tile t-1 in phase
    // A(t)); a wave's DMAs retire in order, so the counted vmcnt before each barrier leaves the younger tile in flight.
    f32x16 S;
    auto qk = [&](int t) {  // S^T[key][q] = K Q^T for this wave's 32 keys of tile t
        const char* Kc = smem + (t & 1) * 2 * TILE;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kc + ((kbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256));
            S = mfma32(a, Qf[ks], S);
        }
        if (t * 64 + 64 > kvl) {  // block-uniform: only the last key tile can be ragged
            const int n0 = t * 64 + 32 * g;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = n0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= kvl) S[i] = -INFINITY;
            }
        }
    };
    auto softmax_pv = [&](int t) {
        const char* Vc = smem + (t & 1) * 2 * TILE + TILE;
        float tmax = S[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, S[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // Lazy rescale: the reference exponent m only follows the running maximum when it has grown by more than 2^8
        // (P then stays below 256: nothing for bf16 / fp32 accumulation), so the D/2-register rescale of O — as much
        // VALU work as the tile's softmax — runs in the first tiles only instead of in most of them (with 32 queries per
        // wave SOME row sets a new maximum in nearly every tile).  m, l and O stay mutually consistent, so the result is
        // the same softmax; the pair merge and the LSE use m as it is.
        const float ts = tmax * sc;
        const float m_new = ts > m + 8.f ? ts : m;
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pv = __builtin_amdgcn_exp2f(S[i] * sc - m_new);
            psum += pv;
            S[i] = pv;
        }
        lsum = lsum * alpha + psum;
        m = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // exact skip once the running maxima have settled
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
        }
        bf16x8 Pf[2];
        Pf[0] = pack8(S, 0);
        Pf[1] = pack8(S, 1);
        // O^T[d][q] += V^T P^T over this wave's 32 keys
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const char* a0 = Vc + kk * 16 * RB + (dt >> 2) * 256;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (vbase[0] ^ ((dt & 3) << 6))));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (vbase[1] ^ ((dt & 3) << 6))));
                s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                O[dt] = mfma32(__builtin_bit_cast(bf16x8, both), Pf[kk], O[dt]);
            }
    };
    constexpr int NI = Stage::NI;
    // Each group issues its share of a DMA where its own S is dead (g0: before its QK^T / after its PV; g1: after its PV /
    // before its QK^T) — S live across an issue costs 16 registers this kernel does not have at head_dim 256.
    auto wait_older = [&](bool two_in_flight) {  // all but this phase's own (younger) tile have landed
        if (two_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (g == 0) {
        for (int t = 0; t < ntiles; ++t) {
            const bool next = t + 1 < ntiles;
            if (next) Stage::issue(rk, smem + ((t + 1) & 1) * 2 * TILE, (t + 1) * 64, kvl, ks_s);
            qk(t);
            wait_older(next);  // V(t)
            __builtin_amdgcn_s_barrier();
            softmax_pv(t);
            if (next) Stage::issue(rv, smem + ((t + 1) & 1) * 2 * TILE + TILE, (t + 1) * 64, kvl, vs_s);
            wait_older(next);  // K(t+1)
            __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int t = 0; t < ntiles; ++t) {
            const bool next = t + 1 < ntiles;
            if (t > 0) softmax_pv(t - 1);
            if (next) Stage::issue(rk, smem + ((t + 1) & 1) * 2 * TILE, (t + 1) * 64, kvl, ks_s);
            wait_older(next);
            __builtin_amdgcn_s_barrier();
            if (next) Stage::issue(rv, smem + ((t + 1) & 1) * 2 * TILE + TILE, (t + 1) * 64, kvl, vs_s);
            qk(t);
            wait_older(next);
            __builtin_amdgcn_s_barrier();
        }
        if (ntiles > 0) softmax_pv(ntiles - 1);
    }
    __syncthreads();  // (g1 leaves the loop one phase after g0: the merge below re-uses the K/V buffers)

    // ---- merge the wave pair: wave g finalises d-tiles [g DT/2, (g+1) DT/2); it receives the partner's accumulators for
    //      those tiles (and the partner's m, l) through the K/V buffers, which are idle now ------------------------------
    constexpr int HT = DT / 2;
    float4* X = reinterpret_cast<float4*>(smem);           // [8 waves][HT*4 float4][64 lanes]
    float2* ML = reinterpret_cast<float2*>(smem + 8 * HT * 4 * 64 * 16);  // [8 waves][64 lanes]
    const float ltot_own = lsum + __shfl_xor(lsum, 32, 64);
    // (register arrays must be indexed by compile-time constants: the two roles are separate code paths)
    auto give = [&](auto GIVE0) {  // write the accumulators of the partner's d-tiles
        constexpr int base = decltype(GIVE0)::value * HT;
#pragma unroll
        for (int j = 0; j < HT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                X[(w * HT * 4 + j * 4 + e) * 64 + lane] =
                    make_float4(O[base + j][4 * e], O[base + j][4 * e + 1], O[base + j][4 * e + 2], O[base + j][4 * e + 3]);
    };
    if (g == 0) give(std::integral_constant<int, 1>{}); else give(std::integral_constant<int, 0>{});
    ML[w * 64 + lane] = make_float2(m, ltot_own);
    __syncthreads();
    const int pw = w ^ 4;
    const float2 pml = ML[pw * 64 + lane];
    const float mm = fmaxf(m, pml.x);
    const float a_own = __builtin_amdgcn_exp2f(m - mm), a_par = __builtin_amdgcn_exp2f(pml.x - mm);
    const float ltot = ltot_own * a_own + pml.y * a_par;
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    auto take = [&](auto MINE) {  // combine, normalise and store this wave's d-tiles
        constexpr int base = decltype(MINE)::value * HT;
#pragma unroll
        for (int j = 0; j < HT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 po = X[(pw * HT * 4 + j * 4 + e) * 64 + lane];
                bf16_t* dst = Ob + (int64_t)qrow * p.so.s + (base + j) * 32 + 8 * e + 4 * h;
                store4(dst, (O[base + j][4 * e] * a_own + po.x * a_par) * inv, (O[base + j][4 * e + 1] * a_own + po.y * a_par) * inv,
                       (O[base + j][4 * e + 2] * a_own + po.z * a_par) * inv, (O[base + j][4 * e + 3] * a_own + po.w * a_par) * inv);
            }
    };
    if (qok) {
        if (g == 0) take(std::integral_constant<int, 0>{}); else take(std::integral_constant<int, 1>{});
        if (g == 0 && h == 0 && p.lse) {
            p.lse[((int64_t)b * p.Hq + hq) * p.Sq + qrow] = ltot > 0.f ? (mm + __log2f(ltot)) * kLn2 : -INFINITY;
        }
    }
}

// =============================================================================================
// backward dQ: one workgroup = 128 queries, key tiles of 64
// =============================================================================================
template <int D>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_bwd_dq_kernel(AttnParams p) {
    constexpr int LD = D + 8, KS = D / 16, DT = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + 64 * LD;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    int qtile, bh;
    if (!xcd_map((p.Sq + 127) / 128, p.B * p.Hq, qtile, bh)) return;  // block-uniform
    const int b = bh / p.Hq, hq = bh % p.Hq, q0 = qtile * 128;
    const int hkv = hq / (p.Hq / p.Hkv);
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Qb = p.q + b * p.sq.b + hq * p.sq.h;
    const bf16_t* dOb = p.dout + b * p.sdo.b + hq * p.sdo.h;
    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dQb = p.dq + b * p.sdq.b + hq * p.sdq.h;

    const int qrow = q0 + w * 32 + r;
    const bool qok = qrow < p.Sq;
    bf16x8 Qf[KS], dOf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Qf[ks] = ldg_frag(Qb + (int64_t)qrow * p.sq.s + ks * 16 + 8 * h, qok);
        dOf[ks] = ldg_frag(dOb + (int64_t)qrow * p.sdo.s + ks * 16 + 8 * h, qok);
    }
    const int64_t srow = ((int64_t)b * p.Hq + hq) * p.Sq + qrow;
    const float lse2 = qok ? p.lse[srow] * kLog2e : INFINITY;
    // delta = rowsum(dO * O), computed here (this kernel runs first) and published for the dK/dV kernel
    float dl = 0.f;
    {
        const bf16_t* Ob = p.o + b * p.so.b + hq * p.so.h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 of = ldg_frag(Ob + (int64_t)qrow * p.so.s + ks * 16 + 8 * h, qok);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)of[j] * (float)dOf[ks][j];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (qok && h == 0) p.delta[srow] = dl;
    }
    const float sc = p.scale * kLog2e;

    f32x16 dQ[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;

    const int ntiles = (kvl + 63) / 64;
    TileStage<D, 64> kst, vst;
    kst.init(Kb, p.sk.s);
    vst.init(Vb, p.sv.s);
    constexpr int BUF = 2 * 64 * LD;  // double-buffered K/V tiles, one barrier per tile (see the forward kernel)
    if (ntiles > 0) {
        kst.load(0, kvl);
        vst.load(0, kvl);
        kst.store(Ks);
        vst.store(Vs);
    }
    __syncthreads();
    if (ntiles > 1) {
        kst.load(64, kvl);
        vst.load(64, kvl);
    }
    const int frag_off = r * LD + 8 * h;
    const int tr_off = tr_lane_off(LD, lane);
    for (int t = 0; t < ntiles; ++t) {
        const bf16_t* Kc = Ks + (t & 1) * BUF;
        const bf16_t* kfrag = Kc + frag_off;
        const bf16_t* vfrag = Kc + 64 * LD + frag_off;
        const bf16_t* ktr = Kc + tr_off;
        const int n0 = t * 64;
        bf16x8 dSf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[i] = 0.f;
                dP[i] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 a = *reinterpret_cast<const bf16x8*>(kfrag + kt * 32 * LD + ks * 16);
                S = mfma32(a, Qf[ks], S);
                bf16x8 c = *reinterpret_cast<const bf16x8*>(vfrag + kt * 32 * LD + ks * 16);
                dP = mfma32(c, dOf[ks], dP);
            }
            if (n0 + 64 > kvl) {  // ragged last tile: masked keys contribute nothing
                asm volatile("; ragged tail tile" ::: "memory");
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = n0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= kvl) S[i] = -INFINITY;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(S[i] * sc - lse2);
                S[i] = pv * (dP[i] - dl);
            }
            dSf[kt * 2] = pack8(S, 0);
            dSf[kt * 2 + 1] = pack8(S, 1);
        }
        if (t + 1 < ntiles) {
            bf16_t* Kn = Ks + ((t + 1) & 1) * BUF;
            kst.store(Kn);
            vst.store(Kn + 64 * LD);
            if (t + 2 < ntiles) {
                kst.load((t + 2) * 64, kvl);
                vst.load((t + 2) * 64, kvl);
            }
        }
        // dQ^T[d][q] += K^T dS^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a = tr_frag(ktr, LD, kk * 16, dt * 32);
                dQ[dt] = mfma32(a, dSf[kk], dQ[dt]);
            }
        __syncthreads();
    }
    if (qok) {
        const float s = p.scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16_t* dst = dQb + (int64_t)qrow * p.sdq.s + dt * 32 + 8 * g + 4 * h;
                store4(dst, dQ[dt][4 * g] * s, dQ[dt][4 * g + 1] * s, dQ[dt][4 * g + 2] * s, dQ[dt][4 * g + 3] * s);
            }
    }
}

// =============================================================================================
// backward dQ, head_dim >= 128, second form: 4 waves x 32 queries, one wave per SIMD (as above: Q / dO fragments and the dQ
// accumulator take D registers, the rest of the 512-register file pays for read-ahead).  K / V tiles of 64 keys arrive by
// LDS-DMA into the swizzled unpadded image, double-buffered (a tile is ~100 MFMAs of work: ample cover for the DMA issued
// at its start); masked key rows read zeros through the descriptor's bounds check.  Row fragments of K and V run PF
// steps ahead of the S / dP MFMAs, the transposed K fragments TF steps ahead of the dQ MFMAs.
// =============================================================================================
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attn_bwd_dq3_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, RB = D * 2, TILE = 64 * RB;
    typedef DmaTile<D, 64, 4, (VFT_ATTN_BWD_ASM_DMA != 0 || D >= 128)> Stage;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qtile, bh;
    if (!xcd_map((p.Sq + 127) / 128, p.B * p.Hq, qtile, bh)) return;  // block-uniform
    const int b = bh / p.Hq, hq = bh % p.Hq, q0 = qtile * 128;
    const int hkv = hq / (p.Hq / p.Hkv);
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Qb = p.q + b * p.sq.b + hq * p.sq.h;
    const bf16_t* dOb = p.dout + b * p.sdo.b + hq * p.sdo.h;
    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dQb = p.dq + b * p.sdq.b + hq * p.sdq.h;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((int64_t)(p.Sk - 1) * p.sk.s + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((int64_t)(p.Sk - 1) * p.sv.s + D) * 2), 0x00020000);
    const int ks_s = (int)p.sk.s, vs_s = (int)p.sv.s;
    const int ntiles = (kvl + 63) / 64;
    if (ntiles > 0) {
        Stage::issue(rk, smem, 0, kvl, ks_s);
        Stage::issue(rv, smem + TILE, 0, kvl, vs_s);
    }

    const int qrow = q0 + w * 32 + r;
    const bool qok = qrow < p.Sq;
    bf16x8 Qf[KS], dOf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Qf[ks] = ldg_frag(Qb + (int64_t)qrow * p.sq.s + ks * 16 + 8 * h, qok);
        dOf[ks] = ldg_frag(dOb + (int64_t)qrow * p.sdo.s + ks * 16 + 8 * h, qok);
    }
    const int64_t srow = ((int64_t)b * p.Hq + hq) * p.Sq + qrow;
    const float lse2 = qok ? p.lse[srow] * kLog2e : INFINITY;
    // delta = rowsum(dO * O), computed here (this kernel runs first) and published for the dK/dV kernel
    float dl = 0.f;
    {
        const bf16_t* Ob = p.o + b * p.so.b + hq * p.so.h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 of = ldg_frag(Ob + (int64_t)qrow * p.so.s + ks * 16 + 8 * h, qok);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)of[j] * (float)dOf[ks][j];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (qok && h == 0) p.delta[srow] = dl;
    }
    const float sc = p.scale * kLog2e;
    float lse2p = lse2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        pin_loaded(Qf[ks]);
        pin_loaded(dOf[ks]);
    }
    pin_loaded(lse2p);
    pin_loaded(dl);

    f32x16 dQ[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment addresses (bytes inside a 64-row tile): row reads: key row 32 kt + r, chunk 2 ks + h; transposed reads:
    // key rows 16 kk + 4 h + (i >> 2) (+8), chunk 4 dt + 2 ((lane >> 4) & 1) + ((i & 3) >> 1)
    const int rbase = r * RB + ((h ^ swz_f<RB>(r)) << 4);
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int tbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 4 * h + (i16 >> 2) + 8 * hi;
        tbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;
    }
    constexpr int PF = 4, TF = 4;

    for (int t = 0; t < ntiles; ++t) {
        const char* Kc = smem + (t & 1) * 2 * TILE;
        const char* Vc = Kc + TILE;
        if (t + 1 < ntiles) {  // the other buffer was last read before the barrier that ended tile t-1
            char* Kn = smem + ((t + 1) & 1) * 2 * TILE;
            Stage::issue(rk, Kn, (t + 1) * 64, kvl, ks_s);
            Stage::issue(rv, Kn + TILE, (t + 1) * 64, kvl, vs_s);
        }
        const int n0 = t * 64;
        auto rowoff = [&](int st) {  // st = kt * KS + ks
            const int kt = st / KS, ks = st % KS;
            return kt * 32 * RB + (rbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256;
        };
        bf16x8 ka[PF], va[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            ka[j] = *reinterpret_cast<const bf16x8*>(Kc + rowoff(j));
            va[j] = *reinterpret_cast<const bf16x8*>(Vc + rowoff(j));
        }
        bf16x8 dSf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[i] = 0.f;
                dP[i] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int st = kt * KS + ks;
                const bf16x8 a = ka[st % PF], c = va[st % PF];
                if (st + PF < 2 * KS) {
                    ka[st % PF] = *reinterpret_cast<const bf16x8*>(Kc + rowoff(st + PF));
                    va[st % PF] = *reinterpret_cast<const bf16x8*>(Vc + rowoff(st + PF));
                }
                S = mfma32(a, Qf[ks], S);
                dP = mfma32(c, dOf[ks], dP);
            }
            if (n0 + 64 > kvl) {  // ragged last tile: masked keys contribute nothing
                asm volatile("; ragged tail tile" ::: "memory");
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = n0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= kvl) S[i] = -INFINITY;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(S[i] * sc - lse2p);
                S[i] = pv * (dP[i] - dl);
            }
            dSf[kt * 2] = pack8(S, 0);
            dSf[kt * 2 + 1] = pack8(S, 1);
        }
        // dQ^T[d][q] += K^T dS^T
        auto trfrag = [&](int st) {  // st = dt * 4 + kk
            const int dt = st >> 2, kk = st & 3;
            const int o0 = kk * 16 * RB + (dt >> 2) * 256, x = (dt & 3) << 6;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Kc + o0 + (tbase[0] ^ x)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Kc + o0 + (tbase[1] ^ x)));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        bf16x8 ta[TF];
#pragma unroll
        for (int j = 0; j < TF; ++j) ta[j] = trfrag(j);
#pragma unroll
        for (int st = 0; st < 4 * DT; ++st) {
            const bf16x8 a = ta[st % TF];
            if (st + TF < 4 * DT) ta[st % TF] = trfrag(st + TF);
            dQ[st >> 2] = mfma32(a, dSf[st & 3], dQ[st >> 2]);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile t+1 has landed (issued a whole tile of compute ago)
        __builtin_amdgcn_s_barrier();
    }
    if (qok) {
        const float s_ = p.scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16_t* dst = dQb + (int64_t)qrow * p.sdq.s + dt * 32 + 8 * e + 4 * h;
                store4(dst, dQ[dt][4 * e] * s_, dQ[dt][4 * e + 1] * s_, dQ[dt][4 * e + 2] * s_, dQ[dt][4 * e + 3] * s_);
            }
    }
}

// =============================================================================================
// backward dK/dV: one workgroup = 128 keys (4 waves x 32 keys), query tiles of BQ rows,
// loops over the query heads that share this kv head
// =============================================================================================
template <int D, int BQ>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_bwd_dkdv_kernel(AttnParams p) {
    constexpr int LD = D + 8, KS = D / 16, DT = D / 32, QT = BQ / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two buffers of [Q tile | dO tile | lse | delta]: the next tile is written while this one is read,
    // one barrier per tile
    constexpr int BUFB = 2 * BQ * LD * (int)sizeof(bf16_t) + 2 * BQ * (int)sizeof(float);  // bytes per buffer

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    int ktile, bh;
    if (!xcd_map(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv, ktile, bh)) return;  // block-uniform
    const int split = ktile % p.nsplit;  // split-Q: this workgroup reduces a slice of the query tiles
    ktile /= p.nsplit;
    const int b = bh / p.Hkv, hkv = bh % p.Hkv, n0 = ktile * 128;
    const int G = p.Hq / p.Hkv;
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dKb = p.dk + b * p.sdk.b + hkv * p.sdk.h;
    bf16_t* dVb = p.dv + b * p.sdv.b + hkv * p.sdv.h;

    const int key = n0 + w * 32 + r;
    const bool kmem = key < p.Sk;   // memory-valid
    const bool kval = key < kvl;    // participates in the softmax

    bf16x8 Kf[KS], Vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Kf[ks] = ldg_frag(Kb + (int64_t)key * p.sk.s + ks * 16 + 8 * h, kval);
        Vf[ks] = ldg_frag(Vb + (int64_t)key * p.sv.s + ks * 16 + 8 * h, kval);
    }
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dK[dt][i] = 0.f;
            dV[dt][i] = 0.f;
        }
    const float sc = p.scale * kLog2e;

    const int nqt = (p.Sq + BQ - 1) / BQ;
    const int niter = (n0 < kvl) ? G * nqt : 0;   // fully masked key blocks get zero gradients
    TileStage<D, BQ> qst, dost;
    float lse_r = INFINITY, dl_r = 0.f;
    int cur_g = -1;
    const int frag_off = r * LD + 8 * h;
    const int tr_off = tr_lane_off(LD, lane);
    auto prefetch = [&](int it) {
        const int g = it / nqt, t = it % nqt;
        const int hq = hkv * G + g;
        if (g != cur_g) {  // block-uniform: next query head of the GQA group
            qst.init(p.q + b * p.sq.b + hq * p.sq.h, p.sq.s);
            dost.init(p.dout + b * p.sdo.b + hq * p.sdo.h, p.sdo.s);
            cur_g = g;
        }
        qst.load(t * BQ, p.Sq);
        dost.load(t * BQ, p.Sq);
        if (tid < BQ) {
            const int q = t * BQ + tid;
            const int64_t srow = ((int64_t)b * p.Hq + hq) * p.Sq + q;
            lse_r = q < p.Sq ? p.lse[srow] * kLog2e : INFINITY;
            dl_r = q < p.Sq ? p.delta[srow] : 0.f;
        }
    };
    auto stash = [&](int buf) {
        char* base = smem + buf * BUFB;
        bf16_t* Qn = reinterpret_cast<bf16_t*>(base);
        qst.store(Qn);
        dost.store(Qn + BQ * LD);
        if (tid < BQ) {
            float* ls = reinterpret_cast<float*>(base + 2 * BQ * LD * sizeof(bf16_t));
            ls[tid] = lse_r;
            ls[BQ + tid] = dl_r;
        }
    };
    const int per_split = (niter + p.nsplit - 1) / p.nsplit;
    const int it0 = split * per_split, it1 = min(niter, it0 + per_split);
    if (it0 < it1) {
        prefetch(it0);
        stash(0);
    }
    __syncthreads();
    if (it0 + 1 < it1) prefetch(it0 + 1);

    for (int it = it0; it < it1; ++it) {
        const char* base = smem + ((it - it0) & 1) * BUFB;
        const bf16_t* Qs = reinterpret_cast<const bf16_t*>(base);
        const bf16_t* dOs = Qs + BQ * LD;
        const float* lse_s = reinterpret_cast<const float*>(base + 2 * BQ * LD * sizeof(bf16_t));
        const float* dl_s = lse_s + BQ;
        const bf16_t* qfrag = Qs + frag_off;
        const bf16_t* dofrag = dOs + frag_off;
        const bf16_t* qtr = Qs + tr_off;
        const bf16_t* dotr = dOs + tr_off;
        if (it + 1 < it1) {  // the other buffer was last read before the barrier that ended tile it-1
            stash((it + 1 - it0) & 1);
            if (it + 2 < it1) prefetch(it + 2);
        }

        // per 32-query sub-tile: S, dP -> P, dS (packed bf16) -> its share of dV / dK right away, so only one
        // sub-tile's fragments are live at a time (register pressure decides the waves per SIMD here)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[i] = 0.f;
                dP[i] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 a = *reinterpret_cast<const bf16x8*>(qfrag + qt * 32 * LD + ks * 16);
                S = mfma32(a, Kf[ks], S);
                bf16x8 c = *reinterpret_cast<const bf16x8*>(dofrag + qt * 32 * LD + ks * 16);
                dP = mfma32(c, Vf[ks], dP);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g4 + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float pv = __builtin_amdgcn_exp2f(S[4 * g4 + i] * sc - l4[i]);
                    if (!kval) pv = 0.f;
                    S[4 * g4 + i] = pv;
                    dP[4 * g4 + i] = pv * (dP[4 * g4 + i] - d4[i]);
                }
            }
            bf16x8 Pf[2], dSf[2];
            Pf[0] = pack8(S, 0);
            Pf[1] = pack8(S, 1);
            dSf[0] = pack8(dP, 0);
            dSf[1] = pack8(dP, 1);
            // dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int kk = 2 * qt + k2;
                    bf16x8 a = tr_frag(dotr, LD, kk * 16, dt * 32);
                    dV[dt] = mfma32(a, Pf[k2], dV[dt]);
                    bf16x8 c = tr_frag(qtr, LD, kk * 16, dt * 32);
                    dK[dt] = mfma32(c, dSf[k2], dK[dt]);
                }
        }
        __syncthreads();
    }
    if (p.nsplit > 1) {
        // fp32 partials, summed in a fixed order by attn_bwd_split_reduce_kernel (deterministic)
        if (kmem) {
            const int64_t plane = (int64_t)p.B * p.Hkv * p.Sk * D;
            float* wk = p.split_ws + (((int64_t)split * p.B * p.Hkv + bh) * p.Sk + key) * D;
            float* wv = wk + (int64_t)p.nsplit * plane;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    f32x4 a, c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = dK[dt][4 * g + i];
                        c[i] = dV[dt][4 * g + i];
                    }
                    *reinterpret_cast<f32x4*>(wk + d) = a;
                    *reinterpret_cast<f32x4*>(wv + d) = c;
                }
        }
        return;
    }
    if (kmem) {
        const float s = p.scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                store4(dKb + (int64_t)key * p.sdk.s + d, dK[dt][4 * g] * s, dK[dt][4 * g + 1] * s,
                       dK[dt][4 * g + 2] * s, dK[dt][4 * g + 3] * s);
                store4(dVb + (int64_t)key * p.sdv.s + d, dV[dt][4 * g], dV[dt][4 * g + 1], dV[dt][4 * g + 2],
                       dV[dt][4 * g + 3]);
            }
    }
}

// =============================================================================================
// backward dK/dV, head_dim >= 128: one workgroup = 128 keys x 8 waves.  A PAIR of waves (w, w ^ 1 — they sit on
// different SIMDs) owns 32 keys; each wave of the pair owns one HALF of the head dimension: its K / V fragments and its
// dK / dV accumulators cover D/2 columns (D/4 + D/2 registers instead of D/2 + D, so two waves fit a SIMD).
// S = Q K^T and dP = dO V^T contract over the whole head dimension: each wave forms the partial sum over its half, the
// pair's "follower" hands its partials (fp32) to the "leader" through LDS, the leader forms P and dS once for the pair
// and hands them back packed (bf16), and each wave accumulates its own half of dV^T += dO^T P and dK^T += Q^T dS — no
// accumulator is shared, the outputs need no merge.  Per query tile (32 queries) that is three phases
//     A: partial S, dP (MFMA)   | B1: leader: sum + softmax (VALU), DMA issue   | B2: dV, dK (MFMA)
// separated by workgroup barriers.  Waves 4-7 run the same program ONE BARRIER BEHIND waves 0-3 and the leader of a
// pair alternates between the groups, so the two waves that share a SIMD are never in the same phase: while one is in
// its VALU phase (or idle as a follower) the other issues MFMAs.  Q / dO tiles arrive by LDS-DMA into the swizzled
// unpadded image of the forward kernel, in a ring of three buffers (a tile is issued 1.5 tiles before its first reader,
// in phase B1, when the slowest group has left the buffer it overwrites); every wave issues the same number of DMAs
// per tile (its share of Q and dO, 8 floats of lse / delta), so one counted vmcnt per tile retires its share of the
// next tile one barrier before the first group reads it.
// =============================================================================================
// the same with the `nt` cache policy: the load bypasses the CU's vector L1 (used for data this CU stored itself earlier
// in the launch — the running-dQ scratch of the one-pass backward)
__device__ __forceinline__ void dma16_nt(__amdgpu_buffer_rsrc_t r, char* l, int voff) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(l));
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dst), "v"(voff), "s"(r));
}
__device__ __forceinline__ void dma4_builtin(__amdgpu_buffer_rsrc_t r, char* l, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)l, 4, voff, 0, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t r, char* l, int voff) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(l));
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dst), "v"(voff), "s"(r));
}

template <int D>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkdv2_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, HK = KS / 2, HT = DT / 2, RB = D * 2, BQ = 32;
    constexpr int TILE = BQ * RB;                  // bytes of one Q (or dO) tile
    constexpr int BUFB = 2 * TILE + 512;           // Q | dO | lse[32] (+ 128 B written by the DMA's upper lanes) | delta[32] (+ 128 B)
    constexpr int XA = 3 * BUFB;                   // follower -> leader: partial S, dP: 4 pairs x 8 float4 x 64 lanes
    constexpr int XB = XA + 4 * 8 * 64 * 16;       // leader -> follower: packed P, dS:   4 pairs x 4 uint4 x 64 lanes
    constexpr int DUMP = XB + 4 * 4 * 64 * 16;     // 256 B per wave: where the filler DMAs of waves 2-7 land
    constexpr int NDMA = 2 * DmaTile<D, BQ, 8, false>::NI + 1;  // DMA instructions per wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), grp = w >> 2, pair = w >> 1, g = w & 1;
    const bool lead = ((w & 1) ^ grp) == 0;        // wave-uniform; one leader per SIMD
    int ktile, bh;
    if (!xcd_map(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv, ktile, bh)) return;  // block-uniform
    const int split = ktile % p.nsplit;
    ktile /= p.nsplit;
    const int b = bh / p.Hkv, hkv = bh % p.Hkv, n0 = ktile * 128;
    const int G = p.Hq / p.Hkv;
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dKb = p.dk + b * p.sdk.b + hkv * p.sdk.h;
    bf16_t* dVb = p.dv + b * p.sdv.b + hkv * p.sdv.h;

    const int key = n0 + pair * 32 + r;
    const bool kmem = key < p.Sk;   // memory-valid
    const bool kval = key < kvl;    // participates in the softmax
    bf16x8 Kh[HK], Vh[HK];
#pragma unroll
    for (int ks = 0; ks < HK; ++ks) {
        Kh[ks] = ldg_frag(Kb + (int64_t)key * p.sk.s + (g * HK + ks) * 16 + 8 * h, kval);
        Vh[ks] = ldg_frag(Vb + (int64_t)key * p.sv.s + (g * HK + ks) * 16 + 8 * h, kval);
    }
#pragma unroll
    for (int ks = 0; ks < HK; ++ks) {
        pin_loaded(Kh[ks]);
        pin_loaded(Vh[ks]);
    }
    f32x16 dK[HT], dV[HT];
#pragma unroll
    for (int dt = 0; dt < HT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dK[dt][i] = 0.f;
            dV[dt][i] = 0.f;
        }
    const float sc = p.scale * kLog2e;

    const int nqt = (p.Sq + BQ - 1) / BQ;
    const int niter = (n0 < kvl) ? G * nqt : 0;   // fully masked key blocks get zero gradients
    const int per_split = (niter + p.nsplit - 1) / p.nsplit;
    const int it0 = split * per_split, it1 = min(niter, it0 + per_split);
    typedef DmaTile<D, BQ, 8, false> Stage;
    const int qs_s = (int)p.sq.s, dos_s = (int)p.sdo.s;
    // this wave's share of tile `it`: Q, dO, and one 4-byte DMA — wave 0 brings the tile's 32 lse values, wave 1 its 32
    // delta values, the other waves issue a filler into a dump slot so that EVERY wave has the same number of DMAs per
    // tile (the counted vmcnt below relies on it).  Queries beyond Sq read 0 everywhere: a zero dO row contributes nothing.
    auto stage = [&](int it, int buf) {
        const int gq = it / nqt, t = it % nqt;
        const int hq = hkv * G + gq;
        const bf16_t* Qh_ = p.q + b * p.sq.b + hq * p.sq.h;
        const bf16_t* dOh_ = p.dout + b * p.sdo.b + hq * p.sdo.h;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Qh_, 0, (int)(((int64_t)(p.Sq - 1) * p.sq.s + D) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dOh_, 0, (int)(((int64_t)(p.Sq - 1) * p.sdo.s + D) * 2), 0x00020000);
        char* base = smem + buf * BUFB;
        Stage::issue(rq, base, t * BQ, p.Sq, qs_s);
        Stage::issue(rdo, base + TILE, t * BQ, p.Sq, dos_s);
        const int64_t srow = ((int64_t)b * p.Hq + hq) * p.Sq;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((w == 0 ? p.lse : p.delta) + srow), 0, p.Sq * 4, 0x00020000);
        char* sdst = w == 0 ? base + 2 * TILE : (w == 1 ? base + 2 * TILE + 256 : smem + DUMP + w * 256);  // wave-uniform
        dma4_builtin(rs, sdst, (w < 2 && lane < 32) ? (t * BQ + lane) * 4 : (int)0x80000000u);
    };
    if (it0 < it1) stage(it0, 0);
    if (it0 + 1 < it1) stage(it0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one barrier behind

    // fragment addresses (bytes inside a 32-row tile): row reads: query row r, chunk 2 ks + h (ks in this wave's half);
    // transposed reads: query rows 16 kk + 4 h + (i >> 2) (+8), chunk 4 dt + 2 ((lane >> 4) & 1) + ((i & 3) >> 1)
    // (the half's first chunk c0 = g D/16: its low 4 bits take part in the swizzle's XOR, the rest is a plain offset)
    const int c0 = g * (D / 16);
    const int rbase = r * RB + ((h ^ swz_f<RB>(r) ^ (c0 & 15)) << 4) + (c0 & ~15) * 16;
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int tbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 4 * h + (i16 >> 2) + 8 * hi;
        tbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr) ^ (c0 & 15)) << 4) + (i16 & 1) * 8 + (c0 & ~15) * 16;
    }
    float4* xa = reinterpret_cast<float4*>(smem + XA) + pair * 8 * 64 + lane;   // [e][lane]
    u32x4* xb = reinterpret_cast<u32x4*>(smem + XB) + pair * 4 * 64 + lane;

    int buf = 0;
    for (int it = it0; it < it1; ++it) {
        const char* Qs = smem + buf * BUFB;
        const char* dOs = Qs + TILE;
        const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * TILE);
        const float* dl_s = lse_s + 64;
        // lane-constant addresses, made opaque per tile: hoisted out of the loop their XOR-ed variants (8 + 8 registers)
        // would stay live across it, and this kernel has none to spare
        int rb = rbase, tb0 = tbase[0], tb1 = tbase[1];
        asm volatile("" : "+v"(rb), "+v"(tb0), "+v"(tb1));
        // ---- phase A: partial S and dP over this wave's half of the head dimension ----
        f32x16 S, dP;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            S[i] = 0.f;
            dP[i] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < HK; ++ks) {
            const int o = (rb ^ ((ks & 7) << 5)) + (ks >> 3) * 256;
            S = mfma32(*reinterpret_cast<const bf16x8*>(Qs + o), Kh[ks], S);
            dP = mfma32(*reinterpret_cast<const bf16x8*>(dOs + o), Vh[ks], dP);
        }
        if (!lead) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xa[e * 64] = make_float4(S[4 * e], S[4 * e + 1], S[4 * e + 2], S[4 * e + 3]);
                xa[(4 + e) * 64] = make_float4(dP[4 * e], dP[4 * e + 1], dP[4 * e + 2], dP[4 * e + 3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase B1: tile it+2 on its way (its buffer was last read in phase B2 of tile it-1, which both groups have
        //      left); the leader sums the partials and forms P, dS for the pair ----
        const bool more2 = it + 2 < it1;
        bf16x8 Pf[2], dSf[2];
        if (lead) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 a = xa[g4 * 64], c = xa[(4 + g4) * 64];
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 8 * g4 + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + 8 * g4 + 4 * h);
                const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float pv = __builtin_amdgcn_exp2f((S[4 * g4 + i] + av[i]) * sc - l4[i] * kLog2e);
                    if (!kval) pv = 0.f;
                    S[4 * g4 + i] = pv;
                    dP[4 * g4 + i] = pv * (dP[4 * g4 + i] + cv[i] - d4[i]);
                }
            }
            Pf[0] = pack8(S, 0);
            Pf[1] = pack8(S, 1);
            dSf[0] = pack8(dP, 0);
            dSf[1] = pack8(dP, 1);
            xb[0] = __builtin_bit_cast(u32x4, Pf[0]);
            xb[64] = __builtin_bit_cast(u32x4, Pf[1]);
            xb[128] = __builtin_bit_cast(u32x4, dSf[0]);
            xb[192] = __builtin_bit_cast(u32x4, dSf[1]);
        }
        if (more2) stage(it + 2, buf == 0 ? 2 : buf - 1);
        // this wave's share of tile it+1 (issued one tile ago) has landed: the group ahead reads it right after the
        // barrier below; only the DMAs just issued for tile it+2 stay in flight
        if (more2) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        // ---- phase B2: dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS   (d in this wave's half) ----
        if (!lead) {
            Pf[0] = __builtin_bit_cast(bf16x8, xb[0]);
            Pf[1] = __builtin_bit_cast(bf16x8, xb[64]);
            dSf[0] = __builtin_bit_cast(bf16x8, xb[128]);
            dSf[1] = __builtin_bit_cast(bf16x8, xb[192]);
        }
#pragma unroll
        for (int dt = 0; dt < HT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int o0 = kk * 16 * RB + (dt >> 2) * 256, x = (dt & 3) << 6;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dOs + o0 + (tb0 ^ x)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dOs + o0 + (tb1 ^ x)));
                dV[dt] = mfma32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), Pf[kk], dV[dt]);
                s16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Qs + o0 + (tb0 ^ x)));
                s16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Qs + o0 + (tb1 ^ x)));
                dK[dt] = mfma32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(lo2, hi2, 0, 1, 2, 3, 4, 5, 6, 7)), dSf[kk], dK[dt]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers
    const int dcol0 = g * (D / 2);
    if (p.nsplit > 1) {
        // fp32 partials, summed in a fixed order by attn_bwd_split_reduce_kernel (deterministic)
        if (kmem) {
            const int64_t plane = (int64_t)p.B * p.Hkv * p.Sk * D;
            float* wk = p.split_ws + (((int64_t)split * p.B * p.Hkv + bh) * p.Sk + key) * D;
            float* wv = wk + (int64_t)p.nsplit * plane;
#pragma unroll
            for (int dt = 0; dt < HT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = dcol0 + dt * 32 + 8 * e + 4 * h;
                    f32x4 a, c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = dK[dt][4 * e + i];
                        c[i] = dV[dt][4 * e + i];
                    }
                    *reinterpret_cast<f32x4*>(wk + d) = a;
                    *reinterpret_cast<f32x4*>(wv + d) = c;
                }
        }
        return;
    }
    if (kmem) {
        const float s_ = p.scale;
#pragma unroll
        for (int dt = 0; dt < HT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = dcol0 + dt * 32 + 8 * e + 4 * h;
                store4(dKb + (int64_t)key * p.sdk.s + d, dK[dt][4 * e] * s_, dK[dt][4 * e + 1] * s_, dK[dt][4 * e + 2] * s_,
                       dK[dt][4 * e + 3] * s_);
                store4(dVb + (int64_t)key * p.sdv.s + d, dV[dt][4 * e], dV[dt][4 * e + 1], dV[dt][4 * e + 2], dV[dt][4 * e + 3]);
            }
    }
}

// =============================================================================================
// backward dK/dV, head_dim >= 128, third form: 4 waves x 32 keys, the whole head dimension per wave, ONE wave per SIMD
// with the full 512-register file — the accumulators (dK, dV: D registers) and the K / V fragments (D/2) leave room for
// what the first-generation kernel lacked: fragment reads issued several steps ahead of the MFMAs that consume them.
// (Measured on that kernel and on the 8-wave split above: every MFMA waited for its own ds_read, one read in flight per
// wave, ~8 200 cycles per 32-query tile for 2 048 cycles of MFMA work, whatever the wave arrangement.)
// Q / dO tiles (32 queries) arrive by LDS-DMA into the swizzled unpadded image of the forward kernel, in a ring of four
// buffers filled three tiles ahead; tiles past the end are issued as all-out-of-range DMAs (zeros, no memory traffic) so
// that the counted vmcnt is the same in every iteration.  One barrier per tile.
// =============================================================================================
template <int D, int BQ, bool STORE_DS = false>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attn_bwd_dkdv3_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, RB = D * 2, NB = 4, QT = BQ / 32;
    constexpr int NST = STORE_DS ? 2 * QT : 0;  // 16-byte dS^T stores per lane and tile (always issued: out-of-range ones are dropped)
    static_assert(!STORE_DS || D >= 128, "stored dS: head_dim >= 128");
    static_assert(BQ == 32 || BQ == 64, "query tile");
    constexpr int TILE = BQ * RB;           // bytes of one Q (or dO) tile
    constexpr int BUFB = 2 * TILE + 512;    // Q | dO | lse[BQ] | delta[BQ]  (256-byte slots: one 4-byte DMA each)
    constexpr int DUMP = NB * BUFB;         // 256 B per wave for the filler DMAs of waves 2, 3
    typedef DmaTile<D, BQ, 4, (VFT_ATTN_BWD_ASM_DMA != 0 || D >= 128)> Stage;
    constexpr int NDMA = 2 * Stage::NI + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ktile, bh;
    if (!xcd_map(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv, ktile, bh)) return;  // block-uniform
    const int split = ktile % p.nsplit;
    ktile /= p.nsplit;
    const int b = bh / p.Hkv, hkv = bh % p.Hkv, n0 = ktile * 128;
    const int G = p.Hq / p.Hkv;
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dKb = p.dk + b * p.sdk.b + hkv * p.sdk.h;
    bf16_t* dVb = p.dv + b * p.sdv.b + hkv * p.sdv.h;

    const int key = n0 + w * 32 + r;
    const bool kmem = key < p.Sk;   // memory-valid
    const bool kval = key < kvl;    // participates in the softmax
    bf16x8 Kf[KS], Vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Kf[ks] = ldg_frag(Kb + (int64_t)key * p.sk.s + ks * 16 + 8 * h, kval);
        Vf[ks] = ldg_frag(Vb + (int64_t)key * p.sv.s + ks * 16 + 8 * h, kval);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        pin_loaded(Kf[ks]);
        pin_loaded(Vf[ks]);
    }
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dK[dt][i] = 0.f;
            dV[dt][i] = 0.f;
        }
    const float sc = p.scale * kLog2e;

    const int nqt = (p.Sq + BQ - 1) / BQ;
    const int niter = (n0 < kvl) ? G * nqt : 0;   // fully masked key blocks get zero gradients
    const int per_split = (niter + p.nsplit - 1) / p.nsplit;
    const int it0 = split * per_split, it1 = min(niter, it0 + per_split);
    const int qs_s = (int)p.sq.s, dos_s = (int)p.sdo.s;
    // this wave's share of tile `it` (all out of range past the end): Q, dO, and one 4-byte DMA — wave 0 the tile's BQ lse
    // values, wave 1 its delta values, waves 2-3 a filler, so that every wave issues NDMA instructions per tile
    // (query head of the kv group, query tile) of the tile staged next: advanced by hand — one wave per SIMD hides nothing,
    // and a scalar division per tile is ~20 dependent instructions in the open
    int sg = it0 / nqt, stl = it0 - sg * nqt;
    typedef DmaTileRel<(D >= 128 ? D : 128), (D >= 128 ? BQ : 32), 4> StageRel;
    // this lane's part of every DMA address (see DmaTileRel): row offset under either row stride, swizzled chunk, stats slot
    int lrow_q = StageRel::lane_row() * qs_s * 2, lrow_do = StageRel::lane_row() * dos_s * 2, lchunk = StageRel::lane_chunk_bytes();
    int lstat = lane < BQ ? lane * 4 : (int)0x80000000u;
    if constexpr (D >= 128) {
        pin_loaded(lrow_q);
        pin_loaded(lrow_do);
        pin_loaded(lchunk);
        pin_loaded(lstat);
    }
    auto stage = [&](int it, int buf) {
        const bool live = it < it1;
        const int gq = live ? sg : 0, t = live ? stl : 0;
        const int hq = hkv * G + gq;
        const bf16_t* Qh_ = p.q + b * p.sq.b + hq * p.sq.h;
        const bf16_t* dOh_ = p.dout + b * p.sdo.b + hq * p.sdo.h;
        char* base = smem + buf * BUFB;
        const int64_t srow = ((int64_t)b * p.Hq + hq) * p.Sq;
        char* sdst = w == 0 ? base + 2 * TILE : (w == 1 ? base + 2 * TILE + 256 : smem + DUMP + w * 256);  // wave-uniform
        if constexpr (D >= 128) {
            // descriptors that start at the tile's first row and end with the head's rows (a dead tile: no bytes at all)
            // (round 5 tried to keep these three as incrementally advanced pointers — ~60 of the loop's 134 scalar instructions
            // are their 64-bit products — but the kernel has no scalar register left: the nine extra SGPRs pushed four lane
            // constants into scratch, and their reload put two `s_waitcnt vmcnt(0)` at the top of every tile: DESIGN §8)
            const int row0 = t * BQ;
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(Qh_ + (int64_t)row0 * p.sq.s), 0, live ? (int)(((int64_t)(p.Sq - 1 - row0) * p.sq.s + D) * 2) : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)(dOh_ + (int64_t)row0 * p.sdo.s), 0, live ? (int)(((int64_t)(p.Sq - 1 - row0) * p.sdo.s + D) * 2) : 0, 0x00020000);
            StageRel::issue(rq, base, lrow_q, lchunk, qs_s * 2);
            StageRel::issue(rdo, base + TILE, lrow_do, lchunk, dos_s * 2);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((w == 0 ? p.lse : p.delta) + srow + row0), 0, (live && w < 2) ? (p.Sq - row0) * 4 : 0, 0x00020000);
            dma4(rs, sdst, lstat);
        } else {
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Qh_, 0, (int)(((int64_t)(p.Sq - 1) * p.sq.s + D) * 2), 0x00020000);
            const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dOh_, 0, (int)(((int64_t)(p.Sq - 1) * p.sdo.s + D) * 2), 0x00020000);
            const int lim = live ? p.Sq : 0;
            Stage::issue(rq, base, t * BQ, lim, qs_s);
            Stage::issue(rdo, base + TILE, t * BQ, lim, dos_s);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((w == 0 ? p.lse : p.delta) + srow), 0, p.Sq * 4, 0x00020000);
            if constexpr (VFT_ATTN_BWD_ASM_DMA != 0) dma4(rs, sdst, (live && w < 2 && lane < BQ) ? (t * BQ + lane) * 4 : (int)0x80000000u);
            else dma4_builtin(rs, sdst, (live && w < 2 && lane < BQ) ? (t * BQ + lane) * 4 : (int)0x80000000u);
        }
        if (++stl == nqt) stl = 0, ++sg;
    };
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) {
        stage(it0 + j, j);
        if constexpr (STORE_DS) {
            // The loop's counted wait allows the dS stores of two tiles in flight behind the DMAs of tile it+1; NST dropped
            // (out-of-range) stores here make that count exact from the first iteration on.
            if (j == NB - 3) {
                const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc((void*)p.ds_t, 0, 0, 0x00020000);
#pragma unroll
                for (int e = 0; e < NST; ++e) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rnull, (int)0x80000000u, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * NDMA + NST) : "memory");  // the first tile has landed
    __builtin_amdgcn_s_barrier();

    // fragment addresses (bytes inside a 32-row tile): row reads: query row r, chunk 2 ks + h;
    // transposed reads: query rows 16 kk + 4 h + (i >> 2) (+8), chunk 4 dt + 2 ((lane >> 4) & 1) + ((i & 3) >> 1)
    const int rbase = r * RB + ((h ^ swz_f<RB>(r)) << 4);
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int tbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 4 * h + (i16 >> 2) + 8 * hi;
        tbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;
    }
    constexpr int PF = D >= 256 ? 3 : 4;                     // row fragments in flight ahead of their MFMAs (what the register file affords)
    constexpr int TF = D >= 256 ? 2 : (2 * DT < 3 ? 2 * DT : 3);  // (d-tile, k-step) steps of transposed fragments in flight
    static_assert(PF <= KS && TF <= 2 * DT, "ring depth");

    // stored dS: (query head, query tile) of the tile being computed, its plane of dS^T, this lane's fixed byte offset in it
    // (rows of ds_ld = 32 * nqt elements: no query guard; keys past the end are out of range for good)
    int cg = it0 / nqt, ct = it0 - cg * nqt;
    __amdgpu_buffer_rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc((void*)p.q, 0, 0, 0x00020000);
    int ds_voff = (int)0x80000000u;
    auto ds_plane = [&]() {
        if constexpr (STORE_DS)
            rds = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ds_t + ((int64_t)b * p.Hq + hkv * G + cg) * p.Sk * p.ds_ld), 0, (int)((int64_t)p.Sk * p.ds_ld * 2), 0x00020000);
    };
    if constexpr (STORE_DS) {
        ds_plane();
        if (kmem) ds_voff = (int)(((int64_t)key * p.ds_ld + 8 * h) * 2);
    }
    int buf = 0;
    for (int it = it0; it < it1; ++it) {
        const char* Qs = smem + buf * BUFB;
        const char* dOs = Qs + TILE;
        const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * TILE);
        const float* dl_s = lse_s + 64;
        stage(it + NB - 1, buf == 0 ? NB - 1 : buf - 1);  // the buffer tile it-1 has just left
        auto rowoff = [&](int ks) { return (rbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256; };
        auto trfrag = [&](const char* tile, int step) {  // step = dt * 2 + kk; tile = the sub-tile's first row
            const int dt = step >> 1, kk = step & 1;
            const int o0 = kk * 16 * RB + (dt >> 2) * 256, x = (dt & 3) << 6;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o0 + (tbase[0] ^ x)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o0 + (tbase[1] ^ x)));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        if constexpr (D >= 128) {
            // One wave per SIMD: nothing but this wave's own instruction stream hides an LDS round trip, and left to itself
            // the compiler serialises read -> wait -> MFMA (round-2 ISA at head_dim 256: 63 waits for 64 MFMAs, 54 of them
            // lgkmcnt(0) / lgkmcnt(1)).  Explicit software pipeline instead: the fragments of CH k-steps are read as one
            // group, two groups in flight (double-buffered registers), scheduling barriers keep a group's reads ahead of
            // the previous group's MFMAs.
            constexpr int CH = D >= 256 ? 2 : 4, NG1 = KS / CH, NG2 = 2 * DT / CH;  // (head_dim 256: groups of 4 spill 24 registers)
            static_assert(KS % CH == 0 && (2 * DT) % CH == 0, "chunking");
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const char* Qq = Qs + qt * 32 * RB;
                const char* dOq = dOs + qt * 32 * RB;
                bf16x8 qa[2][CH], da[2][CH];
                auto load1 = [&](int g) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        qa[g & 1][j] = *reinterpret_cast<const bf16x8*>(Qq + rowoff(g * CH + j));
                        da[g & 1][j] = *reinterpret_cast<const bf16x8*>(dOq + rowoff(g * CH + j));
                    }
                };
                f32x16 S, dP;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[i] = 0.f;
                    dP[i] = 0.f;
                }
                load1(0);
#pragma unroll
                for (int g = 0; g < NG1; ++g) {
                    if (g + 1 < NG1) load1(g + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        S = mfma32(qa[g & 1][j], Kf[g * CH + j], S);
                        dP = mfma32(da[g & 1][j], Vf[g * CH + j], dP);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                bf16x8 ta[2][CH], tc[2][CH];
                auto load2 = [&](int g) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        ta[g & 1][j] = trfrag(dOq, g * CH + j);
                        tc[g & 1][j] = trfrag(Qq, g * CH + j);
                    }
                };
                load2(0);  // travels while the softmax runs on the VALU
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g4 + 4 * h);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float pv = __builtin_amdgcn_exp2f(S[4 * g4 + i] * sc - l4[i] * kLog2e);
                        if (!kval) pv = 0.f;  // (round 5: hoisting this select into a branch taken only by ragged key blocks, as in
                                              //  attn_bwd1_kernel, costs this kernel a 4-register scratch reload — and with it an
                                              //  `s_waitcnt vmcnt(0)` — at the top of every tile: it has no register to spare)
                        S[4 * g4 + i] = pv;
                        dP[4 * g4 + i] = pv * (dP[4 * g4 + i] - d4[i]);
                    }
                }
                bf16x8 Pf[2], dSf[2];
                Pf[0] = pack8(S, 0);
                Pf[1] = pack8(S, 1);
                dSf[0] = pack8(dP, 0);
                dSf[1] = pack8(dP, 1);
#pragma unroll
                for (int g = 0; g < NG2; ++g) {
                    if (g + 1 < NG2) load2(g + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int st = g * CH + j;
                        dV[st >> 1] = mfma32(ta[g & 1][j], Pf[st & 1], dV[st >> 1]);
                        dK[st >> 1] = mfma32(tc[g & 1][j], dSf[st & 1], dK[st >> 1]);
                    }
                    if constexpr (STORE_DS) {
                        // dS^T[key][q] for the batched dQ GEMM (vft_wg8_batched), in the shadow of this group's MFMAs.  A lane
                        // holds, of its key, queries 8 g + 4 h + i: swapping the upper half-wave of the even-g registers with the
                        // lower half-wave of the odd-g ones (v_permlane32_swap) leaves 8 consecutive queries = one 16-byte
                        // store per lane and 16-query pair.  Lane part of the address fixed, tile part in the scalar offset.
                        if (g < 2) {
                            const u32x4 d = __builtin_bit_cast(u32x4, dSf[g]);
                            const auto x0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                            const auto x1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                            const u32x4 o = {x0[0], x1[0], x0[1], x1[1]};
                            __builtin_amdgcn_raw_buffer_store_b128(o, rds, ds_voff, (ct * BQ + qt * 32 + 16 * g) * 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
    #pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const char* Qq = Qs + qt * 32 * RB;
                const char* dOq = dOs + qt * 32 * RB;
                // ---- S = Q K^T, dP = dO V^T: fragment reads run PF steps ahead ----
                bf16x8 qa[PF], da[PF];
    #pragma unroll
                for (int j = 0; j < PF; ++j) {
                    qa[j] = *reinterpret_cast<const bf16x8*>(Qq + rowoff(j));
                    da[j] = *reinterpret_cast<const bf16x8*>(dOq + rowoff(j));
                }
                f32x16 S, dP;
    #pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[i] = 0.f;
                    dP[i] = 0.f;
                }
    #pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 a = qa[ks % PF], c = da[ks % PF];
                    if (ks + PF < KS) {
                        qa[ks % PF] = *reinterpret_cast<const bf16x8*>(Qq + rowoff(ks + PF));
                        da[ks % PF] = *reinterpret_cast<const bf16x8*>(dOq + rowoff(ks + PF));
                    }
                    S = mfma32(a, Kf[ks], S);
                    dP = mfma32(c, Vf[ks], dP);
                }
                // the first transposed fragments of the second half travel while the softmax runs on the VALU
                bf16x8 ta[TF], tc[TF];
    #pragma unroll
                for (int j = 0; j < TF; ++j) {
                    ta[j] = trfrag(dOq, j);
                    tc[j] = trfrag(Qq, j);
                }
    #pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g4 + 4 * h);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 32 + 8 * g4 + 4 * h);
    #pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float pv = __builtin_amdgcn_exp2f(S[4 * g4 + i] * sc - l4[i] * kLog2e);
                        if (!kval) pv = 0.f;
                        S[4 * g4 + i] = pv;
                        dP[4 * g4 + i] = pv * (dP[4 * g4 + i] - d4[i]);
                    }
                }
                bf16x8 Pf[2], dSf[2];
                Pf[0] = pack8(S, 0);
                Pf[1] = pack8(S, 1);
                dSf[0] = pack8(dP, 0);
                dSf[1] = pack8(dP, 1);
                // ---- dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS ----
    #pragma unroll
                for (int st = 0; st < 2 * DT; ++st) {
                    const bf16x8 a = ta[st % TF], c = tc[st % TF];
                    if (st + TF < 2 * DT) {
                        ta[st % TF] = trfrag(dOq, st + TF);
                        tc[st % TF] = trfrag(Qq, st + TF);
                    }
                    dV[st >> 1] = mfma32(a, Pf[st & 1], dV[st >> 1]);
                    dK[st >> 1] = mfma32(c, dSf[st & 1], dK[st >> 1]);
                }
            }
        }
        // this wave's share of tile it+1 has landed (issued three tiles ago); tiles it+2, it+3 — and the dS stores of tiles
        // it-1, it, issued after them — stay in flight (vector memory operations retire in issue order)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NB - 2) * (NDMA + NST)) : "memory");
        __builtin_amdgcn_s_barrier();
        buf = buf == NB - 1 ? 0 : buf + 1;
        if constexpr (STORE_DS) {
            if (++ct == nqt) {
                ct = 0, ++cg;
                ds_plane();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of ours may still write LDS when the workgroup ends
    if (p.nsplit > 1) {
        // fp32 partials, summed in a fixed order by attn_bwd_split_reduce_kernel (deterministic)
        if (kmem) {
            const int64_t plane = (int64_t)p.B * p.Hkv * p.Sk * D;
            float* wk = p.split_ws + (((int64_t)split * p.B * p.Hkv + bh) * p.Sk + key) * D;
            float* wv = wk + (int64_t)p.nsplit * plane;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = dt * 32 + 8 * e + 4 * h;
                    f32x4 a, c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = dK[dt][4 * e + i];
                        c[i] = dV[dt][4 * e + i];
                    }
                    *reinterpret_cast<f32x4*>(wk + d) = a;
                    *reinterpret_cast<f32x4*>(wv + d) = c;
                }
        }
        return;
    }
    if (kmem) {
        const float s_ = p.scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = dt * 32 + 8 * e + 4 * h;
                store4(dKb + (int64_t)key * p.sdk.s + d, dK[dt][4 * e] * s_, dK[dt][4 * e + 1] * s_, dK[dt][4 * e + 2] * s_,
                       dK[dt][4 * e + 3] * s_);
                store4(dVb + (int64_t)key * p.sdv.s + d, dV[dt][4 * e], dV[dt][4 * e + 1], dV[dt][4 * e + 2], dV[dt][4 * e + 3]);
            }
    }
}

// =============================================================================================
// backward dK/dV, head_dim 96 (Lumina2: 24 query / 8 kv heads x 96), round 3: 4 waves x 32 keys, 64-query tiles, two
// workgroups per CU (two waves per SIMD that do not share barriers).  The round-1 kernel for this head_dim staged Q / dO
// through registers (24 VGPRs held across the tile), spilled inside the loop and — as its ISA showed — waited lgkmcnt(0)
// in front of 50 of its 48 MFMAs' operand reads: one LDS round trip per MFMA.  This one:
//   * Q / dO tiles by asm LDS-DMA into rows PADDED to 256 bytes (the 192-byte rows of head_dim 96 do not divide a DMA piece;
//     with 256-byte rows the swizzled image of the head_dim 128 / 256 kernels applies unchanged; pad chunks are requested
//     out of range and arrive as zeros), double-buffered, one barrier per tile;
//   * the explicit stage pipeline of the one-pass kernel (R: row fragments, M1: S / dP, T: transposed fragments in flight
//     under the softmax, M2: dV / dK; the next sub-tile's row fragments issued before this one's M2).
// GQA: loops over the query heads of the kv head; short key sequences split their query tiles (nsplit) like the others.
// =============================================================================================
template <int D, int ROWS, int NW>
struct DmaTilePad {  // ROWS x D bf16 -> LDS rows of 256 bytes (D * 2 <= 256), swizzle on the source chunk
    static constexpr int NI = ROWS * 256 / 1024 / NW;
    static_assert(D * 2 <= 256 && ROWS * 256 % (1024 * NW) == 0, "geometry");
    static __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, char* lds_tile, int row0, int row_limit, int row_stride_elems) {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        int voffs[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int byte = (wv * NI + q) * 1024 + lane * 16;
            const int r = byte >> 8, slot = (byte & 255) >> 4;
            const int c = slot ^ swz_f<256>(r);
            voffs[q] = (row0 + r < row_limit && c < D / 8) ? (row0 + r) * row_stride_elems * 2 + c * 16 : (int)0x80000000u;
        }
        dma16_group<NI>(rs, lds_tile + wv * NI * 1024, voffs);
    }
};

template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv4_kernel(AttnParams p) {
    constexpr int KS = D / 16, DT = D / 32, RB = 256, BQ = 64, QT = 2;
    constexpr int TILE = BQ * RB;            // bytes of one Q (or dO) tile in LDS
    constexpr int BUFB = 2 * TILE + 512;     // Q | dO | lse[64] | delta[64]
    constexpr int DUMP = 2 * BUFB;           // 256 B per wave for the filler DMAs of waves 2, 3
    typedef DmaTilePad<D, BQ, 4> Stage;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ktile, bh;
    if (!xcd_map(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv, ktile, bh)) return;  // block-uniform
    const int split = ktile % p.nsplit;
    ktile /= p.nsplit;
    const int b = bh / p.Hkv, hkv = bh % p.Hkv, n0 = ktile * 128;
    const int G = p.Hq / p.Hkv;
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);

    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dKb = p.dk + b * p.sdk.b + hkv * p.sdk.h;
    bf16_t* dVb = p.dv + b * p.sdv.b + hkv * p.sdv.h;
    const int key = n0 + w * 32 + r;
    const bool kmem = key < p.Sk, kval = key < kvl;
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dK[dt][i] = 0.f;
            dV[dt][i] = 0.f;
        }
    const float sc = p.scale * kLog2e;
    const int nqt = (p.Sq + BQ - 1) / BQ;
    const int niter = (n0 < kvl) ? G * nqt : 0;
    const int per_split = (niter + p.nsplit - 1) / p.nsplit;
    const int it0 = split * per_split, it1 = min(niter, it0 + per_split);
    const int qs_s = (int)p.sq.s, dos_s = (int)p.sdo.s;
    auto stage = [&](int it, int buf) {  // this wave's share of tile `it` (nothing real past the end): Q, dO, one 4-byte DMA
        const bool live = it < it1;
        const int itc = live ? it : it0;
        const int gq = itc / nqt, t = itc % nqt;
        const int hq = hkv * G + gq;
        const bf16_t* Qh_ = p.q + b * p.sq.b + hq * p.sq.h;
        const bf16_t* dOh_ = p.dout + b * p.sdo.b + hq * p.sdo.h;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Qh_, 0, (int)(((int64_t)(p.Sq - 1) * p.sq.s + D) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dOh_, 0, (int)(((int64_t)(p.Sq - 1) * p.sdo.s + D) * 2), 0x00020000);
        char* base = smem + buf * BUFB;
        const int lim = live ? p.Sq : 0;
        Stage::issue(rq, base, t * BQ, lim, qs_s);
        Stage::issue(rdo, base + TILE, t * BQ, lim, dos_s);
        const int64_t srow = ((int64_t)b * p.Hq + hq) * p.Sq;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((w == 0 ? p.lse : p.delta) + srow), 0, p.Sq * 4, 0x00020000);
        char* sdst = w == 0 ? base + 2 * TILE : (w == 1 ? base + 2 * TILE + 256 : smem + DUMP + w * 256);  // wave-uniform
        dma4(rs, sdst, (live && w < 2) ? (t * BQ + lane) * 4 : (int)0x80000000u);
    };
    if (it0 < it1) stage(it0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int rbase = r * RB + ((h ^ swz_f<RB>(r)) << 4);
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int tbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 4 * h + (i16 >> 2) + 8 * hi;
        tbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;
    }
    // The K / V fragments are loaded HERE, behind the staging prologue, and waited for before the loop.  Round 3 loaded them
    // first; the prologue's address arithmetic then spilled all 12, hipcc reloaded them in front of the loop and carried the
    // reloads into it as pending: the first 12 MFMAs of EVERY iteration waited vmcnt(11) .. vmcnt(0) — on the hardware counter
    // those are the nine DMA pieces of the next tile issued a moment earlier, so the prefetch distance was an eighth of a
    // tile instead of a tile (seen in the ISA, round 4).
    bf16x8 Kf[KS], Vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Kf[ks] = ldg_frag(Kb + (int64_t)key * p.sk.s + ks * 16 + 8 * h, kval);
        Vf[ks] = ldg_frag(Vb + (int64_t)key * p.sv.s + ks * 16 + 8 * h, kval);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        pin_loaded(Kf[ks]);
        pin_loaded(Vf[ks]);
    }
    int buf = 0;
    for (int it = it0; it < it1; ++it) {
        const char* Qs = smem + buf * BUFB;
        const char* dOs = Qs + TILE;
        const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * TILE);
        const float* dl_s = lse_s + 64;
        stage(it + 1, buf ^ 1);  // the other buffer: every wave left it at the barrier that ended tile it-1
        // lane-constant addresses, made opaque per use site: hoisted out of the loop their XOR-ed variants (6 row + 6 transposed
        // addresses) stay live across it — and this kernel has no register to spare (a spill inside the loop is a scratch load,
        // i.e. a vmcnt(0) that drains the DMA)
        int rb = rbase, tb0 = tbase[0], tb1 = tbase[1];
        auto rowoff = [&](int ks) { return (rb ^ ((ks & 7) << 5)) + (ks >> 3) * 256; };
        auto trfrag = [&](const char* tile, int dt) {  // 32 columns 32 dt.., 16 rows from `tile`
            const int x = (dt & 3) << 6;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + (dt >> 2) * 256 + (tb0 ^ x)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + (dt >> 2) * 256 + (tb1 ^ x)));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        bf16x8 qa[KS], da[KS], ta[2 * DT], tc[2 * DT];
        auto read_rows = [&](int qt) {
            const char* Qq = Qs + qt * 32 * RB;
            const char* dOq = dOs + qt * 32 * RB;
            asm volatile("" : "+v"(rb));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qa[ks] = *reinterpret_cast<const bf16x8*>(Qq + rowoff(ks));
                da[ks] = *reinterpret_cast<const bf16x8*>(dOq + rowoff(ks));
            }
        };
        read_rows(0);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const char* Qq = Qs + qt * 32 * RB;
            const char* dOq = dOs + qt * 32 * RB;
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[i] = 0.f;
                dP[i] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {  // M1
                S = mfma32(qa[ks], Kf[ks], S);
                dP = mfma32(da[ks], Vf[ks], dP);
            }
            __builtin_amdgcn_sched_barrier(0);  // (T strictly behind M1: its registers are the row fragments')
            asm volatile("" : "+v"(tb0), "+v"(tb1));
#pragma unroll
            for (int st = 0; st < 2 * DT; ++st) {  // T: in flight while the softmax runs
                ta[st] = trfrag(dOq + (st & 1) * 16 * RB, st >> 1);
                tc[st] = trfrag(Qq + (st & 1) * 16 * RB, st >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {  // V
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g4 + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float pv = __builtin_amdgcn_exp2f(S[4 * g4 + i] * sc - l4[i] * kLog2e);
                    if (!kval) pv = 0.f;
                    S[4 * g4 + i] = pv;
                    dP[4 * g4 + i] = pv * (dP[4 * g4 + i] - d4[i]);
                }
            }
            bf16x8 Pf[2], dSf[2];
            Pf[0] = pack8(S, 0);
            Pf[1] = pack8(S, 1);
            dSf[0] = pack8(dP, 0);
            dSf[1] = pack8(dP, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2 * DT; ++st) {  // M2
                dV[st >> 1] = mfma32(ta[st], Pf[st & 1], dV[st >> 1]);
                dK[st >> 1] = mfma32(tc[st], dSf[st & 1], dK[st >> 1]);
            }
            // (issued behind the MFMAs, not before them: at head_dim 96 row + transposed fragments together do not fit the 256
            // registers of two waves per SIMD; the reads overlap the matrix pipe's drain and the other workgroup's wave)
            if (qt + 1 < QT) read_rows(qt + 1);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile it+1 has landed; this wave's reads of tile it are done
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.nsplit > 1) {
        if (kmem) {
            const int64_t plane = (int64_t)p.B * p.Hkv * p.Sk * D;
            float* wk = p.split_ws + (((int64_t)split * p.B * p.Hkv + bh) * p.Sk + key) * D;
            float* wv = wk + (int64_t)p.nsplit * plane;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = dt * 32 + 8 * e + 4 * h;
                    f32x4 a, c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = dK[dt][4 * e + i];
                        c[i] = dV[dt][4 * e + i];
                    }
                    *reinterpret_cast<f32x4*>(wk + d) = a;
                    *reinterpret_cast<f32x4*>(wv + d) = c;
                }
        }
        return;
    }
    if (kmem) {
        const float s_ = p.scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = dt * 32 + 8 * e + 4 * h;
                store4(dKb + (int64_t)key * p.sdk.s + d, dK[dt][4 * e] * s_, dK[dt][4 * e + 1] * s_, dK[dt][4 * e + 2] * s_,
                       dK[dt][4 * e + 3] * s_);
                store4(dVb + (int64_t)key * p.sdv.s + d, dV[dt][4 * e], dV[dt][4 * e + 1], dV[dt][4 * e + 2], dV[dt][4 * e + 3]);
            }
    }
}

// =============================================================================================
// backward in ONE pass (head_dim 64): dQ, dK and dV from a single sweep — five MFMA products per (query, key) block
// instead of the seven of the dq + dkdv pair (S and dP are formed once), and one softmax / dS evaluation instead of two.
//
// One workgroup = 8 waves = 256 keys of one (batch, kv head): wave w keeps K / V fragments and the dK^T / dV^T
// accumulators of its 32 keys in registers (two waves per SIMD) and the workgroup sweeps the query tiles (64 queries) of
// the query heads that share the kv head, exactly like the dK/dV kernel above.  What is new is dQ:
//   * S and dP are computed with the KEY on the MFMA lane (that is what makes P / dS the B operands of the dV / dK
//     products), so dS^T — the operand dQ needs, keys in the contraction — does not exist in registers.  Each wave
//     writes its dS block (bf16) into a [256 keys][64 queries] LDS image: ONE pass through LDS, four 8-byte stores per
//     32 x 32 block, in the swizzled layout that serves transposed reads.
//   * dQ^T[d][q] = sum_keys K^T[d][key] dS^T[key][q]: both operands are transposed reads (ds_read_b64_tr_b16) — of a K
//     image staged once per key block and of the dS image.  The 64 x 64 output is four 32 x 32 blocks; wave w takes
//     block (w & 3) over the key half (w >> 2) — 8 MFMAs per wave per tile on top of the 32 of S, dP, dV, dK — and the
//     two waves of a block (w, w ^ 4: one SIMD) swap half of their accumulators through LDS, so each ends up with 8
//     complete fp32 rows of the block's sum over the workgroup's 256 keys.
//   * The sum over KEY BLOCKS needs no atomics and no second workgroup: the workgroup walks its key blocks ONE AFTER THE
//     OTHER (a (batch, head) pair — or a contiguous slice of its key blocks when there are too few pairs to fill the chip
//     — belongs to one workgroup) and keeps the running dQ in an fp32 scratch plane that only it touches:
//     read-add-write per tile, the same lane re-reading what it stored one key-block pass earlier (loads bypass L1),
//     values prefetched one tile ahead.  Fixed order => bit-reproducible.  The first block skips the read, the last one
//     writes bf16 dQ (scaled) straight to the output; with a sliced key range the planes of the slices are added, in
//     slice order, by attn_bwd1_reduce_kernel.
// Cost: the scratch read-modify-write (32 bytes per (query, d) element and key block = 1.25 * 256 flop per byte) and
// the Q / dO re-read per key block; at 256 keys per block that is ~2.4 + 1.2 TB/s beside an MFMA-bound loop.
// Q / dO tiles arrive by LDS-DMA in a ring of three buffers; two workgroup barriers per tile.
// =============================================================================================
// Round 5: the per-tile bookkeeping on the scalar unit is incremental.  The ISA of the round-3/4 loop
// carried 226 scalar and 213 vector instructions per 64-query tile and wave beside its 40 MFMAs — three integer divisions
// (tile -> (query head, query tile) for the staging, the scratch plane and the final store), four buffer descriptors rebuilt
// from 64-bit products, per-lane staging addresses and range checks recomputed per tile, a per-element key-mask select —
// and at two waves per SIMD the kernel is bound by instruction ISSUE (~640 instructions x ~4.5 cycles per wave and tile
// against 1 280 cycles of MFMA work), not by the matrix pipe, LDS or HBM.  Now: cursors advance by one tile, descriptors
// are rebuilt only when the query head changes, the tensors' ends are the descriptors' ends (rows past them read zeros by
// the hardware range check), the scratch plane of a (batch, kv head) pair is ONE descriptor indexed by the tile number, and
// the key mask is applied only in a ragged key block: 653 -> ~460 instructions per tile and wave.  Measured (A/B of the two
// forms in one process, gpurun_out/r5d): 3.872 -> 3.734 ms at B24 H10 S4096, 0.580 -> 0.564 ms at B24 H20 S1024 — a 30 %
// shorter instruction stream buys 3-4 %: the kernel is NOT bound by instruction issue.  What it is bound by, from timing
// runs with parts switched off (tools/exp/attn_bwd1_probe.py, build with -DVFT_ATTN_PROBE; S4096: 3.82 ms whole; 3.36
// without the running-dQ traffic; 3.10 without the Q / dO staging traffic; 2.96 without both; 2.73 with the dQ product cut
// to an eighth as well): ~0.9 ms of memory stalls (12.4 GB per launch at 3.2 TB/s, planes + Q / dO of 240 workgroups that
// share nothing), and a 2.7 ms core in which the eight waves run M1 -> softmax -> M2 in lock step between two workgroup
// barriers per tile (both waves of a SIMD want the matrix pipe, then both the vector ALU): DESIGN §10.
#ifdef VFT_ATTN_STAMP  // diagnostic build only (tools/exp/attn_bwd1_stamps.py): cycles per segment of the tile loop, summed per wave of workgroup 0
__device__ unsigned long long g_bwd1_stamps[8][16];
#define BWD1_STAMP(k)                                                                  \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        unsigned long long t_;                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        seg[k] += (unsigned)(t_ - tprev);                                              \
        tprev = t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
extern "C" int vft_attn_probe_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd1_stamps), sizeof(g_bwd1_stamps));
}
#else
#define BWD1_STAMP(k)
#endif
template <int D>
__global__ __launch_bounds__(512, 2) void attn_bwd1_kernel(AttnParams p) {
#ifdef VFT_ATTN_STAMP
    unsigned seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
#endif
#ifdef VFT_ATTN_PROBE
    const int dbg = p.dbg;
    // (bits 16 / 32: static issue priority for waves 4-7 / for the odd waves — measured: no effect)
    if ((dbg & 16) && (threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);
    if ((dbg & 32) && ((threadIdx.x >> 6) & 1)) __builtin_amdgcn_s_setprio(1);
#else
    constexpr int dbg = 0;
#endif
    static_assert(D == 64, "the one-pass backward is built for head_dim 64 (register budget at two waves per SIMD)");
    constexpr int KS = D / 16, DT = D / 32, RB = D * 2, BQ = 64, QT = BQ / 32, KB = 256, NB = 3;
    constexpr int TILE = BQ * RB;                 // bytes of one Q (or dO) tile
    constexpr int BUFB = 2 * TILE + 512;          // Q | dO | lse[64] | delta[64]
    constexpr int KIMG = NB * BUFB;               // K image of the key block: [256][D] bf16, swizzled
    constexpr int DSIMG = KIMG + KB * RB;         // dS image: [256 keys][64 queries] bf16 (128-byte rows), swizzled
    constexpr int RBS = BQ * 2;                   // bytes per dS row
    constexpr int XCH = DSIMG + KB * RBS;         // pair exchange: 8 waves x 2 float4 x 64 lanes
    constexpr int DUMP = XCH + 8 * 2048;          // 256 B per wave for the filler DMAs of waves 2-7
    constexpr int OLD = DUMP + 8 * 256;           // running dQ of the next tile: 2 KiB per wave, by LDS-DMA from the scratch plane
    typedef DmaTile<D, BQ, 8> Stage;
    static_assert(Stage::NI == 1 && RBS == RB, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkb = (p.Sk + KB - 1) / KB;
    const int bh = blockIdx.x / p.nsplit, split = blockIdx.x % p.nsplit;
    if (bh >= p.B * p.Hkv) return;                // block-uniform
    const int b = bh / p.Hkv, hkv = bh % p.Hkv;
    const int G = p.Hq / p.Hkv;
    int kvl = p.Sk;
    if (p.kv_len) kvl = min(max(p.kv_len[b], 0), p.Sk);
    const int per_split = (nkb + p.nsplit - 1) / p.nsplit;
    const int kb0 = split * per_split, kb1 = min(nkb, kb0 + per_split);

    const bf16_t* Kb = p.k + b * p.sk.b + hkv * p.sk.h;
    const bf16_t* Vb = p.v + b * p.sv.b + hkv * p.sv.h;
    bf16_t* dKb = p.dk + b * p.sdk.b + hkv * p.sdk.h;
    bf16_t* dVb = p.dv + b * p.sdv.b + hkv * p.sdv.h;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((int64_t)(p.Sk - 1) * p.sk.s + D) * 2), 0x00020000);
    const float sc = p.scale * kLog2e;
    const int nqt = (p.Sq + BQ - 1) / BQ;
    const int niter = G * nqt;
    const int qs_s = (int)p.sq.s, dos_s = (int)p.sdo.s, ks_s = (int)p.sk.s, sdq_s = (int)p.sdq.s;
    const int64_t plane = (int64_t)p.B * p.Hq * nqt * (BQ * D);   // floats per scratch plane
    float* const scr = p.split_ws + (int64_t)split * plane;

    // ---- staging state: the cursor (query head of the kv group, query tile) of the next tile to stage, the descriptors
    //      of that head (Q, dO from the head's first row to its last; -lse/scale or -delta for waves 0 / 1, empty for the
    //      others), and this lane's fixed part of the addresses (Stage: one 1 KiB piece per wave = rows 8 w .. 8 w + 7)
    int sg = 0, stl = 0;
    const int st_r = w * 8 + (lane >> 3);
    const int st_c = ((lane & 7) ^ swz_f<RB>(st_r)) << 4;
    int vq_lane = st_r * qs_s * 2 + st_c, vdo_lane = st_r * dos_s * 2 + st_c, vst_lane = lane * 4;
    __amdgpu_buffer_rsrc_t hrq = __builtin_amdgcn_make_buffer_rsrc((void*)p.q, 0, 0, 0x00020000), hrdo = hrq, hrs = hrq;
    auto head_rsrc = [&](int gq) {  // (gq >= G: past the last head — empty descriptors, every DMA of a dead tile reads zeros)
        const bool livehead = gq < G && !(dbg & 8);
        const int hq = hkv * G + (gq < G ? gq : 0);
        hrq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q + b * p.sq.b + hq * p.sq.h), 0, livehead ? (int)(((int64_t)(p.Sq - 1) * p.sq.s + D) * 2) : 0, 0x00020000);
        hrdo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dout + b * p.sdo.b + hq * p.sdo.h), 0, livehead ? (int)(((int64_t)(p.Sq - 1) * p.sdo.s + D) * 2) : 0, 0x00020000);
        hrs = __builtin_amdgcn_make_buffer_rsrc((void*)((w == 0 ? p.nlse : p.delta) + ((int64_t)b * p.Hq + hq) * p.Sq), 0, (livehead && w < 2) ? p.Sq * 4 : 0, 0x00020000);
    };
    auto stage_lean = [&](int buf) {  // stages the tile under the cursor, then advances it
        char* base = smem + buf * BUFB;
        const int row0 = stl * BQ;
        {
            const int v0[1] = {vq_lane + row0 * qs_s * 2};
            dma16_group<1>(hrq, base + w * 1024, v0);
            const int v1[1] = {vdo_lane + row0 * dos_s * 2};
            dma16_group<1>(hrdo, base + TILE + w * 1024, v1);
        }
        char* sdst = w == 0 ? base + 2 * TILE : (w == 1 ? base + 2 * TILE + 256 : smem + DUMP + w * 256);  // wave-uniform
        dma4(hrs, sdst, vst_lane + row0 * 4);
        if (++stl == nqt) {
            stl = 0;
            ++sg;
            head_rsrc(sg);
        }
    };
    // this wave's share of tile `it` (all out of range past the end): Q, dO, and one 4-byte DMA — wave 0 the tile's 64
    // -lse / scale values, wave 1 its -delta values (both prepared by attn_delta_kernel), the others a filler
    auto stage = [&](int /*it*/, int buf) { stage_lean(buf); };

    // fragment addresses (bytes inside a tile of 128-byte rows): row reads: row r, chunk 2 ks + h; transposed reads: rows
    // 16 kk + 4 h + (i >> 2) (+8), chunk 4 cb + 2 ((lane >> 4) & 1) + ((i & 3) >> 1)
    const int rbase = r * RB + ((h ^ swz_f<RB>(r)) << 4);
    const int i16 = lane & 15, cc = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    int tbase[2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int vr = 4 * h + (i16 >> 2) + 8 * hi;
        tbase[hi] = vr * RB + ((cc ^ swz_f<RB>(vr)) << 4) + (i16 & 1) * 8;
    }
    // where this lane's dS values go: row = its key, 8-byte piece (16-byte chunk c, half h) of the swizzled 128-byte row
    const int dsrow = (w * 32 + r) * RBS + 8 * h, dsswz = swz_f<RBS>(w * 32 + r);
    // the dQ block of this wave: d columns 32 (w & 1), queries 32 ((w >> 1) & 1), key half (w >> 2)
    const int dtw = w & 1, qtw = (w >> 1) & 1, kh = w >> 2;
    f32x4* const xch = reinterpret_cast<f32x4*>(smem + XCH);

    for (int kb = kb0; kb < kb1; ++kb) {
        const int n0 = kb * KB;
        const bool first = kb == kb0, last = kb + 1 == kb1;
        const bool final_store = last && p.nsplit == 1;
        const int key = n0 + w * 32 + r;
        const bool kmem = key < p.Sk;   // memory-valid
        const bool kval = key < kvl;    // participates in the softmax
        const bool ragged = n0 + KB > kvl;  // (block-uniform)
        // ---- the key block's operands: K / V fragments of this wave's 32 keys, the K image for the dQ product, the first
        //      two query tiles.  (Every wave has passed the last reads of the previous block: barrier at the loop's end.)
        bf16x8 Kf[KS], Vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Kf[ks] = ldg_frag(Kb + (int64_t)key * p.sk.s + ks * 16 + 8 * h, kval);
            Vf[ks] = ldg_frag(Vb + (int64_t)key * p.sv.s + ks * 16 + 8 * h, kval);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pin_loaded(Kf[ks]);
            pin_loaded(Vf[ks]);
        }
        DmaTile<D, KB, 8>::issue(rk, smem + KIMG, n0, kvl, ks_s);
        sg = 0, stl = 0;
        head_rsrc(0);
        stage(0, 0);
        stage(1, 1);
        f32x16 dK[DT], dV[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                dK[dt][i] = 0.f;
                dV[dt][i] = 0.f;
            }
        auto tile_id_of = [&](int it) {
            const int gq = it / nqt, t = it % nqt;
            return ((int64_t)b * p.Hq + hkv * G + gq) * nqt + t;
        };
        // The running dQ of tile `it` travels scratch plane -> LDS by DMA one tile ahead (no VGPR-destination load in the
        // loop: beside hidden DMAs the compiler would wait vmcnt(0) for it and drain the ring).  Nothing to read in the
        // range's first key block or past the last tile: an out-of-range offset delivers zeros.
        // the tiles of a (batch, kv head) pair are consecutive in the plane (tile id = first tile of the pair + it): one
        // descriptor over the pair's niter tiles, the tile number times 16 KiB in the lane offset; past the last tile the
        // offset is past the descriptor (zeros); in the range's first key block there is nothing to read (empty descriptor)
        float* const pair_scr = scr + tile_id_of(0) * (BQ * D);
        const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc((void*)pair_scr, 0, (first || (dbg & 1)) ? 0 : niter * (BQ * D * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t wpl = __builtin_amdgcn_make_buffer_rsrc((void*)pair_scr, 0, (dbg & 1) ? 0 : niter * (BQ * D * 4), 0x00020000);
        const int old_lane = ((w & 3) * 1024 + (2 * kh) * 256 + lane * 4) * 4;
        // dQ of the query head under the cursor (last key block of an unsliced range only)
        __amdgpu_buffer_rsrc_t hrdq = rpl;
        auto dq_rsrc = [&](int gq) {
            const bool live = gq < G;
            hrdq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dq + b * p.sdq.b + (hkv * G + (live ? gq : 0)) * p.sdq.h), 0,
                                                     live ? (int)(((int64_t)(p.Sq - 1) * p.sdq.s + D) * 2) : 0, 0x00020000);
        };
        if (final_store) dq_rsrc(0);
        const int dq_lane = ((qtw * 32 + r) * sdq_s + dtw * 32 + 16 * kh + 4 * h) * 2;
        auto old_dma = [&](int it) {
            const int voff = old_lane + it * (BQ * D * 4);
            dma16_nt(rpl, smem + OLD + w * 2048, voff);
            dma16_nt(rpl, smem + OLD + w * 2048 + 1024, voff + 1024);
        };
        old_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        int buf = 0;
        int cg = 0, ct = 0;  // (query head of the kv group, query tile) of the tile being computed
        for (int it = 0; it < niter; ++it) {
            const char* Qs = smem + buf * BUFB;
            const char* dOs = Qs + TILE;
            const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * TILE);
            const float* dl_s = lse_s + 64;
            auto rowoff = [&](int ks) { return (rbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256; };
            auto trfrag = [&](const char* tile, int cb) {  // 32 columns 32 cb.., 16 rows from `tile`
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + (tbase[0] ^ (cb << 6))));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + (tbase[1] ^ (cb << 6))));
                return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            };
            // ---- phase 1: S, dP -> P, dS; dV, dK; dS into the LDS image.  Written as an explicit software pipeline over the
            //      two 32-query sub-tiles (R: row fragments + row constants, M1: S / dP, T: transposed fragments, V: softmax on
            //      the VALU, W: dS stores, M2: dV / dK) with scheduling barriers between the stages: left to itself the
            //      compiler re-uses one fragment register set and every MFMA waits for its own LDS read.
            //      Row constants as the INITIAL ACCUMULATORS: S starts from -lse / scale and dP from -delta, so that
            //      p = exp2(sc * S) and dS = p * dP need no per-element subtraction (the accumulator register index is the
            //      query, so the constants are four 16-byte LDS reads per accumulator).
            bf16x8 qa[KS], da[KS], ta[2 * DT], tc[2 * DT];
            f32x16 S, dP;
            auto read_rows = [&](int qt) {  // R
                const char* Qq = Qs + qt * 32 * RB;
                const char* dOq = dOs + qt * 32 * RB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    qa[ks] = *reinterpret_cast<const bf16x8*>(Qq + rowoff(ks));
                    da[ks] = *reinterpret_cast<const bf16x8*>(dOq + rowoff(ks));
                }
            };
            auto init_acc = [&](int qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g4 + 4 * h);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        S[4 * g4 + i] = l4[i];
                        dP[4 * g4 + i] = d4[i];
                    }
                }
            };
#ifdef VFT_ATTN_STAMP
            {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            read_rows(0);
            init_acc(0);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const char* Qq = Qs + qt * 32 * RB;
                const char* dOq = dOs + qt * 32 * RB;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {  // M1
                    S = mfma32(qa[ks], Kf[ks], S);
                    dP = mfma32(da[ks], Vf[ks], dP);
                }
#pragma unroll
                for (int st = 0; st < 2 * DT; ++st) {  // T: in flight while the softmax runs
                    ta[st] = trfrag(dOq + (st & 1) * 16 * RB, st >> 1);
                    tc[st] = trfrag(Qq + (st & 1) * 16 * RB, st >> 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                BWD1_STAMP(0 + 3 * qt);  // R (first sub-tile) + M1 issue + T issue (+ the stamp's wait for T)
#pragma unroll
                for (int i = 0; i < 16; ++i) {  // V
                    const float pv = __builtin_amdgcn_exp2f(S[i] * sc);
                    S[i] = pv;
                    dP[i] = pv * dP[i];
                }
                if (ragged) {  // block-uniform: only the last key block of a masked / non-multiple key range has keys to drop
                    asm volatile("; ragged key block" ::: "memory");  // (keeps hipcc from if-converting the branch back into 64 selects per tile)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        S[i] = kval ? S[i] : 0.f;
                        dP[i] = kval ? dP[i] : 0.f;
                    }
                }
                bf16x8 Pf[2], dSf[2];
                Pf[0] = pack8(S, 0);
                Pf[1] = pack8(S, 1);
                dSf[0] = pack8(dP, 0);
                dSf[1] = pack8(dP, 1);
                // W: dS block -> image: registers 8 s + 4 e .. + 3 are queries 32 qt + 16 s + 8 e + 4 h .. + 3 of this lane's key
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const u32x4 v = __builtin_bit_cast(u32x4, dSf[s_]);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = qt * 4 + 2 * s_ + e;
                        *reinterpret_cast<u32x2*>(smem + DSIMG + dsrow + (((c ^ dsswz) & 7) << 4)) = u32x2{v[2 * e], v[2 * e + 1]};
                    }
                }
                if (qt + 1 < QT) read_rows(qt + 1);  // the next sub-tile's row fragments fly during this one's dV / dK MFMAs
                __builtin_amdgcn_sched_barrier(0);
                BWD1_STAMP(1 + 3 * qt);  // wait for S / dP + V + W
                // M2: dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS
#pragma unroll
                for (int st = 0; st < 2 * DT; ++st) {
                    dV[st >> 1] = mfma32(ta[st], Pf[st & 1], dV[st >> 1]);
                    dK[st >> 1] = mfma32(tc[st], dSf[st & 1], dK[st >> 1]);
                }
                if (qt + 1 < QT) init_acc(qt + 1);  // (issued behind the MFMAs: S / dP are free again, the reads overlap their execution)
                BWD1_STAMP(2 + 3 * qt);  // M2 issue
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // B1: the dS image is complete; tile it-1's buffer is free
            BWD1_STAMP(6);  // B1
            stage(it + 2, buf == 0 ? NB - 1 : buf - 1);
            BWD1_STAMP(7);  // staging DMAs issued
            // ---- phase 2: dQ^T block of this wave over its key half, operand fragments four steps ahead --------------
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            {
                const char* kim = smem + KIMG + kh * 128 * RB;
                const char* sim = smem + DSIMG + kh * 128 * RBS;
                constexpr int QF = 4;
                bf16x8 ka[QF], sa[QF];
#pragma unroll
                for (int j = 0; j < QF; ++j) {
                    ka[j] = trfrag(kim + j * 16 * RB, dtw);
                    sa[j] = trfrag(sim + j * 16 * RBS, qtw);
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if ((dbg & 2) && kk >= 1) break;
                    const bf16x8 a_ = ka[kk % QF], b_ = sa[kk % QF];
                    if (kk + QF < 8) {
                        ka[kk % QF] = trfrag(kim + (kk + QF) * 16 * RB, dtw);
                        sa[kk % QF] = trfrag(sim + (kk + QF) * 16 * RBS, qtw);
                    }
                    acc = mfma32(a_, b_, acc);
                }
            }
            BWD1_STAMP(8);  // dQ product: transposed reads + 8 MFMAs issued
            // swap halves with the partner wave (w ^ 4, the same block over the other key half): keep quads 2 kh, 2 kh + 1
            f32x4* xo = xch + (w * 2) * 64 + lane;
            if (kh == 0) {
                xo[0] = f32x4{acc[8], acc[9], acc[10], acc[11]};
                xo[64] = f32x4{acc[12], acc[13], acc[14], acc[15]};
            } else {
                xo[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
                xo[64] = f32x4{acc[4], acc[5], acc[6], acc[7]};
            }
            // everything this wave issued before this tile's stage() has landed or retired: the next tile's Q / dO share
            // (issued one tile ago), this tile's running dQ (issued after the previous B2), last tile's stores; only the
            // three DMAs just issued may still fly
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3) : "memory");
            __builtin_amdgcn_s_barrier();  // B2
            BWD1_STAMP(9);  // exchange stores (wait for the dQ MFMAs) + vmcnt + B2
            const f32x4* xi = xch + ((w ^ 4) * 2) * 64 + lane;
            const f32x4* oi = reinterpret_cast<const f32x4*>(smem + OLD + w * 2048) + lane;
            const f32x4 p0 = xi[0], p1 = xi[64];
            const f32x4 o0 = oi[0], o1 = oi[64];
            f32x4 sum0, sum1;
            if (kh == 0) {
                sum0 = f32x4{acc[0], acc[1], acc[2], acc[3]} + p0 + o0;
                sum1 = f32x4{acc[4], acc[5], acc[6], acc[7]} + p1 + o1;
            } else {
                sum0 = f32x4{acc[8], acc[9], acc[10], acc[11]} + p0 + o0;
                sum1 = f32x4{acc[12], acc[13], acc[14], acc[15]} + p1 + o1;
            }
            // stores through buffer descriptors (a per-lane 32-bit offset that never changes + a scalar tile offset; rows past
            // the tensor's end dropped by the range check): no 64-bit per-lane pointers to keep or rebuild
            if (final_store) {
                const float s_ = p.scale;
                bf16x4 v0, v1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v0[i] = (bf16_t)(sum0[i] * s_);
                    v1[i] = (bf16_t)(sum1[i] * s_);
                }
                const int soff = ct * (BQ * sdq_s * 2);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v0), hrdq, dq_lane, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v1), hrdq, dq_lane + 16, soff, 0);
                if (++ct == nqt) {
                    ct = 0;
                    ++cg;
                    dq_rsrc(cg);
                }
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum0), wpl, old_lane, it * (BQ * D * 4), 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum1), wpl, old_lane + 1024, it * (BQ * D * 4), 0);
            }
            old_dma(it + 1);  // (the reads of this tile's copy above are complete: their values went into the sums)
            buf = buf == NB - 1 ? 0 : buf + 1;
            BWD1_STAMP(10);  // read-modify-write of the running dQ + next tile's plane DMA
        }
        // ---- dK, dV of this key block ----
        if (kmem) {
            const float s_ = p.scale;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = dt * 32 + 8 * e + 4 * h;
                    store4(dKb + (int64_t)key * p.sdk.s + d, dK[dt][4 * e] * s_, dK[dt][4 * e + 1] * s_, dK[dt][4 * e + 2] * s_,
                           dK[dt][4 * e + 3] * s_);
                    store4(dVb + (int64_t)key * p.sdv.s + d, dV[dt][4 * e], dV[dt][4 * e + 1], dV[dt][4 * e + 2], dV[dt][4 * e + 3]);
                }
        }
        // nothing of this block may still be in flight towards LDS, and every wave is past its last reads of the K image /
        // the ring, before the next block's prologue overwrites them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#ifdef VFT_ATTN_STAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 12; ++k) g_bwd1_stamps[w][k] = seg[k];
#endif
}

// dQ = scale * (sum over the key-range slices' scratch planes, in slice order) -> bf16; one thread = one float4 of the
// kernel's native layout [tile][block][quad][lane]
template <int D>
__global__ __launch_bounds__(256) void attn_bwd1_reduce_kernel(AttnParams p) {
    constexpr int BQ = 64;
    const int nqt = (p.Sq + BQ - 1) / BQ;
    const int64_t plane = (int64_t)p.B * p.Hq * nqt * (BQ * D);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e * 4 >= plane) return;
    const int lane = (int)(e & 63), quad = (int)((e >> 6) & 3), blk = (int)((e >> 8) & 3);
    const int64_t tile_id = e >> 10;
    const int t = (int)(tile_id % nqt);
    const int bhq = (int)(tile_id / nqt), b = bhq / p.Hq, hq = bhq % p.Hq;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.nsplit; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(p.split_ws + (int64_t)s * plane + e * 4);
        a.x += x.x, a.y += x.y, a.z += x.z, a.w += x.w;
    }
    const int q = t * BQ + (blk >> 1) * 32 + (lane & 31);
    if (q >= p.Sq) return;
    const int d = (blk & 1) * 32 + 8 * quad + 4 * (lane >> 5);
    store4(p.dq + b * p.sdq.b + hq * p.sdq.h + (int64_t)q * p.sdq.s + d, a.x * p.scale, a.y * p.scale, a.z * p.scale, a.w * p.scale);
}

// Row constants of the one-pass backward: delta[b, h, q] = -sum_d dO * O and nlse = -lse / scale (the initial dP / S
// accumulators); 8 lanes per query row.  (The dq kernel of the two-kernel form computes its delta on the side.)
template <int D, bool ONEPASS = true>  // !ONEPASS: +delta only (the stored-dS form: what the dq kernel used to leave for dK/dV)
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
    constexpr int LPR = D / 8;  // lanes per row (16 bytes each)
    const int64_t rows = (int64_t)p.B * p.Hq * p.Sq;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / LPR;
    const int c = (int)(idx % LPR) * 8;
    float dl = 0.f;
    if (row < rows) {
        const int q = (int)(row % p.Sq);
        const int bhq = (int)(row / p.Sq), b = bhq / p.Hq, hq = bhq % p.Hq;
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(p.o + b * p.so.b + hq * p.so.h + (int64_t)q * p.so.s + c);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(p.dout + b * p.sdo.b + hq * p.sdo.h + (int64_t)q * p.sdo.s + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)o[j] * (float)g[j];
    }
#pragma unroll
    for (int o_ = 1; o_ < LPR; o_ <<= 1) dl += __shfl_xor(dl, o_, 64);
    if (row < rows && c == 0) {
        if constexpr (ONEPASS) {
            p.delta[row] = -dl;
            p.nlse[row] = -p.lse[row] / p.scale;
        } else {
            p.delta[row] = dl;
        }
    }
}

// dk / dv = Σ_split partial (fixed order), dk scaled; one thread = 4 consecutive d of one key
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_split_reduce_kernel(AttnParams p) {
    const int64_t rows = (int64_t)p.B * p.Hkv * p.Sk;
    const int64_t plane = rows * D;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * (D / 4)) return;
    const int64_t row = e / (D / 4);
    const int d = (int)(e % (D / 4)) * 4;
    const int key = (int)(row % p.Sk);
    const int bh = (int)(row / p.Sk), b = bh / p.Hkv, hkv = bh % p.Hkv;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.nsplit; ++s) {
        const float* wk = p.split_ws + (int64_t)s * plane + row * D + d;
        const f32x4 x = *reinterpret_cast<const f32x4*>(wk);
        const f32x4 y = *reinterpret_cast<const f32x4*>(wk + (int64_t)p.nsplit * plane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] += x[i];
            c[i] += y[i];
        }
    }
    store4(p.dk + b * p.sdk.b + hkv * p.sdk.h + (int64_t)key * p.sdk.s + d, a[0] * p.scale, a[1] * p.scale, a[2] * p.scale,
           a[3] * p.scale);
    store4(p.dv + b * p.sdv.b + hkv * p.sdv.h + (int64_t)key * p.sdv.s + d, c[0], c[1], c[2], c[3]);
}

// Short key sequences (cross-attention: 77 text tokens) give the dK/dV kernel one workgroup per (batch, head)
// — far fewer than 256 CUs — each walking every query tile.  Split the query tiles over `nsplit` workgroups.
__host__ inline int dkdv_nsplit(int B, int Hq, int Hkv, int Sq, int Sk, int D) {
    const int BQ = D == 256 ? 32 : 64;
    const int64_t blocks = (int64_t)((Sk + 127) / 128) * B * Hkv;
    const int64_t niter = (int64_t)(Hq / Hkv) * ((Sq + BQ - 1) / BQ);
    const int64_t slots = 256 * (D <= 96 ? 2 : 1);  // workgroups the chip holds at once (launch bounds above)
    if (blocks < 256) {  // too few workgroups to fill the chip
        int64_t ns = (512 + blocks - 1) / blocks;
        if (ns > niter / 4) ns = niter / 4;
        if (ns > 32) ns = 32;
        return ns < 1 ? 1 : (int)ns;
    }
    // enough workgroups, but a short last round can waste a large share of the chip (e.g. Lumina2: 1088 workgroups
    // on 512 slots = 2.1 rounds): split while the quantisation loss is above 12 % and the partial sums stay cheap
    auto eff = [&](int64_t ns) {
        const double rounds = (double)(blocks * ns) / (double)slots;
        return rounds / (double)((blocks * ns + slots - 1) / slots);
    };
    int best = 1;
    for (int ns = 2; ns <= 4 && eff(best) < 0.88; ++ns)
        if (niter / ns >= 16 && eff(ns) > eff(best) + 0.05) best = ns;
    return best;
}

// One-pass backward (attn_bwd1_kernel): head_dim 64 and more than one 128-key block of keys (short key sequences — the
// 77-token cross-attention — would fill 30 % of a 256-key block: they stay on the dq + dkdv pair).  Returns the number of
// key-range slices per (batch, kv head), 0 when the kernel does not apply.  A workgroup costs one pass per key block of its
// slice; the chip holds 256 workgroups: fewest rounds x passes wins, ties go to fewer slices (no reduce pass at 1).
__host__ inline int bwd1_slices(int B, int Hq, int Hkv, int Sq, int Sk, int D) {
    (void)Hq, (void)Sq;
    if (D != 64 || Sk <= 128 || g_attn_variant != 0) return 0;
    const int64_t bhk = (int64_t)B * Hkv, nkb = (Sk + 255) / 256;
    // time ~ rounds of the 256 resident workgroups x key-block passes per workgroup.  Slicing (ns > 1) costs one fp32 plane
    // per slice and a reduce pass over all of them (measured: 16 planes of 252 MB = 0.7 ms beside a 3 ms kernel), so it
    // has to buy at least a quarter of the time, and never more than 4 slices.
    auto cost = [&](int64_t ns) { return ((bhk * ns + 255) / 256) * ((nkb + ns - 1) / ns); };
    int best = 1;
    for (int ns = 2; ns <= nkb && ns <= 4; ++ns)
        if (4 * cost(ns) <= 3 * cost(best)) best = ns;
    return best;
}
__host__ inline int64_t bwd1_plane_floats(int B, int Hq, int Sq, int D) { return (int64_t)B * Hq * ((Sq + 63) / 64) * (64 * D); }

// Stored-dS backward (head_dim >= 128, no key mask): the dK/dV kernel already computes dS for every (key, query) pair — it
// writes it out (bf16, transposed) and dQ = scale * dS K becomes one batched GEMM on the weight-gradient kernel instead of a
// second flash kernel that recomputes S and dP: 5 products per pair instead of 7, for one 2-byte write + read per pair.
__host__ inline bool stored_ds_applies(int B, int Hq, int Sq, int Sk, int D) {
    if (D < 128 || g_attn_variant != 0 || Sq % 8 != 0 || Sq < 512 || Sk < 512) return false;
    return (int64_t)Sk * (Sq + 32) * 2 < ((int64_t)1 << 31) - (1 << 20) && (int64_t)B * Hq * Sk * (Sq + 32) * 2 <= ((int64_t)16 << 30);
}
__host__ inline int64_t stored_ds_ld(int Sq) { return (Sq + 31) / 32 * 32; }  // whole 32-query tiles: the kernel's stores need no query guard
__host__ inline int64_t stored_ds_floats(int B, int Hq, int Sq, int Sk) { return (int64_t)B * Hq * Sk * stored_ds_ld(Sq) / 2; }

template <int D>
int launch_bwd1(AttnParams p, hipStream_t st) {
    constexpr size_t lds = 3 * (2 * 64 * D * 2 + 512) + 256 * D * 2 + 256 * 128 + 8 * 2048 + 8 * 256 + 8 * 2048;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd1_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int64_t rows = (int64_t)p.B * p.Hq * p.Sq * (D / 8);
    hipLaunchKernelGGL(attn_delta_kernel<D>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, p);
    VFT_CHECK_LAUNCH("vft_attn_bwd(delta)");
#ifdef VFT_ATTN_PROBE
    {
        const char* e = getenv("VFT_ATTN_DEBUG");
        p.dbg = e ? atoi(e) : 0;
    }
#endif
    hipLaunchKernelGGL(attn_bwd1_kernel<D>, dim3((unsigned)(p.B * p.Hkv * p.nsplit)), dim3(512), lds, st, p);
    VFT_CHECK_LAUNCH("vft_attn_bwd(one pass)");
    if (p.nsplit > 1) {
        const int64_t n = bwd1_plane_floats(p.B, p.Hq, p.Sq, D) / 4;
        hipLaunchKernelGGL(attn_bwd1_reduce_kernel<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
        VFT_CHECK_LAUNCH("vft_attn_bwd(dq reduce)");
    }
    return 0;
}

int check_strides(const char* name, const int64_t* s) {
    for (int i = 0; i < 3; ++i) {
        if (s[i] < 0 || (s[i] % 8) != 0) {
            vft_set_error("vft_attn: %s stride[%d]=%lld must be a non-negative multiple of 8 elements", name, i,
                          (long long)s[i]);
            return 1;
        }
    }
    return 0;
}
int check_ptr(const char* name, const void* p) {
    if (p == nullptr) {
        vft_set_error("vft_attn: %s is null", name);
        return 1;
    }
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) {
        vft_set_error("vft_attn: %s must be 16-byte aligned", name);
        return 1;
    }
    return 0;
}
Str3 mk(const int64_t* s) { return Str3{s[0], s[1], s[2]}; }

int check_dims(int B, int Hq, int Hkv, int Sq, int Sk, int D) {
    VFT_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Sq > 0 && Sk > 0, "vft_attn: non-positive dimension");
    VFT_CHECK_ARG(Hq % Hkv == 0, "vft_attn: Hq=%d must be a multiple of Hkv=%d", Hq, Hkv);
    VFT_CHECK_ARG(D == 64 || D == 96 || D == 128 || D == 256, "vft_attn: head_dim %d unsupported (64, 96, 128, 256)", D);
    VFT_CHECK_ARG((int64_t)B * Hq * ((Sq + 127) / 128 + (Sk + 127) / 128) < (1ll << 30), "vft_attn: problem exceeds grid limits");
    return 0;
}

template <int D>
int launch_fwd(const AttnParams& p, hipStream_t st) {
    constexpr size_t lds = 2 * 2 * 64 * (D + 8) * sizeof(bf16_t);  // double-buffered K + V tiles
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr = true;
    }
    dim3 grid(xcd_grid((p.Sq + 127) / 128, p.B * p.Hq));
    hipLaunchKernelGGL(attn_fwd_kernel<D>, grid, dim3(256), lds, st, p);
    VFT_CHECK_LAUNCH("vft_attn_fwd");
    return 0;
}

template <int D>
int launch_fwd3(const AttnParams& p, hipStream_t st) {  // head_dim 64 / 96: LDS-DMA operand path (round 4)
    constexpr size_t lds = 2 * 2 * 64 * (D == 96 ? 256 : D * 2);  // double-buffered K + V tiles, head_dim 96 rows padded to 256 B
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd3_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid(xcd_grid((p.Sq + 127) / 128, p.B * p.Hq));
    hipLaunchKernelGGL(attn_fwd3_kernel<D>, grid, dim3(256), lds, st, p);
    VFT_CHECK_LAUNCH("vft_attn_fwd");
    return 0;
}

template <int D>
int launch_fwd2(const AttnParams& p, hipStream_t st) {
    // two K/V buffers (they also carry the pair merge: 8 x D/4 float4 x 64 lanes = the same bytes) + the pair's m / l
    constexpr size_t lds = 2 * 2 * 64 * D * sizeof(bf16_t) + 8 * 64 * sizeof(float2);
    static_assert(8 * (D / 64) * 4 * 64 * 16 <= 2 * 2 * 64 * D * 2, "merge area");
    // head_dim 128 runs the staggered schedule (+14 %), head_dim 256 the lockstep one (the staggered body spills there)
    auto* kern = D == 128 ? attn_fwd2s_kernel<D> : attn_fwd2_kernel<D>;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid(xcd_grid((p.Sq + 127) / 128, p.B * p.Hq));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
    VFT_CHECK_LAUNCH("vft_attn_fwd");
    return 0;
}

template <int D, int BQ>
int launch_bwd(const AttnParams& p, hipStream_t st) {
    const int variant = g_attn_variant == 3 ? 0 : g_attn_variant;  // 3 = "the pair where the one-pass / stored-dS forms would run": the default pair
    if constexpr (D >= 128) {
        if (p.ds_t != nullptr) {
            constexpr int BQ3 = 32;
            constexpr size_t lds = 4 * (2 * BQ3 * D * 2 + 512) + 4 * 256;
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv3_kernel<D, BQ3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr = true;
            }
            const int64_t rows = (int64_t)p.B * p.Hq * p.Sq * (D / 8);
            hipLaunchKernelGGL((attn_delta_kernel<D, false>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(delta)");
            dim3 grid(xcd_grid(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv));
            hipLaunchKernelGGL((attn_bwd_dkdv3_kernel<D, BQ3, true>), grid, dim3(256), lds, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(dkdv + dS)");
            if (p.nsplit > 1) {
                const int64_t n = (int64_t)p.B * p.Hkv * p.Sk * (D / 4);
                hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
                VFT_CHECK_LAUNCH("vft_attn_bwd(split reduce)");
            }
            Wg8Batch bt{};
            bt.nbatch = p.B * p.Hq, bt.inner = p.Hq, bt.G = p.Hq / p.Hkv;
            bt.a_bstride = (int64_t)p.Sk * p.ds_ld, bt.b_outer = p.sk.b, bt.b_inner = p.sk.h, bt.o_outer = p.sdq.b, bt.o_inner = p.sdq.h;
            bt.scale = p.scale;
            return vft_wg8_batched(p.ds_t, p.k, p.dq, p.dq_ws, p.Sk, p.Sq, D, p.ds_ld, p.sk.s, p.sdq.s, bt, st);
        }
    }
    if (D != 96 && variant != 1) {
        if constexpr (D != 96) {
            constexpr size_t lds = 2 * 2 * 64 * D * sizeof(bf16_t);  // double-buffered K | V tiles, unpadded
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq3_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr = true;
            }
            dim3 grid(xcd_grid((p.Sq + 127) / 128, p.B * p.Hq));
            hipLaunchKernelGGL(attn_bwd_dq3_kernel<D>, grid, dim3(256), lds, st, p);  // also writes delta
            VFT_CHECK_LAUNCH("vft_attn_bwd(dq)");
        }
    } else {
        constexpr size_t lds = 2 * 2 * 64 * (D + 8) * sizeof(bf16_t);
        static bool attr = false;
        if (!attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<D>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        dim3 grid(xcd_grid((p.Sq + 127) / 128, p.B * p.Hq));
        hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, grid, dim3(256), lds, st, p);  // also writes delta
        VFT_CHECK_LAUNCH("vft_attn_bwd(dq)");
    }
    if (D != 96 && variant == 0) {
        if constexpr (D != 96) {
            constexpr int BQ3 = D <= 64 ? 64 : 32;
            constexpr size_t lds = 4 * (2 * BQ3 * D * 2 + 512) + 4 * 256;  // four Q|dO|stats buffers + filler slots
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv3_kernel<D, BQ3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr = true;
            }
            dim3 grid(xcd_grid(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv));
            hipLaunchKernelGGL((attn_bwd_dkdv3_kernel<D, BQ3>), grid, dim3(256), lds, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(dkdv)");
        }
        if (p.nsplit > 1) {
            const int64_t n = (int64_t)p.B * p.Hkv * p.Sk * (D / 4);
            hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(split reduce)");
        }
    } else if (D >= 128 && variant == 2) {
        if constexpr (D >= 128) {
            // three Q|dO|stats buffers + the pairs' exchange slabs (partials, packed P/dS) + one dump slot per wave
            constexpr size_t lds = 3 * (2 * 32 * D * 2 + 512) + 4 * 8 * 64 * 16 + 4 * 4 * 64 * 16 + 8 * 256;
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv2_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr = true;
            }
            dim3 grid(xcd_grid(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv));
            hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<D>), grid, dim3(512), lds, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(dkdv)");
        }
        if (p.nsplit > 1) {
            const int64_t n = (int64_t)p.B * p.Hkv * p.Sk * (D / 4);
            hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(split reduce)");
        }
    } else if (D == 96 && variant == 0) {
        if constexpr (D == 96) {
            constexpr size_t lds = 2 * (2 * 64 * 256 + 512) + 4 * 256;  // two Q | dO | stats buffers (256-byte rows) + filler slots
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv4_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr = true;
            }
            dim3 grid(xcd_grid(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv));
            hipLaunchKernelGGL((attn_bwd_dkdv4_kernel<D>), grid, dim3(256), lds, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(dkdv)");
        }
        if (p.nsplit > 1) {
            const int64_t n = (int64_t)p.B * p.Hkv * p.Sk * (D / 4);
            hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(split reduce)");
        }
    } else {
        constexpr size_t lds = 2 * (2 * BQ * (D + 8) * sizeof(bf16_t) + 2 * BQ * sizeof(float));  // double-buffered
        static bool attr = false;
        if (!attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<D, BQ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        dim3 grid(xcd_grid(((p.Sk + 127) / 128) * p.nsplit, p.B * p.Hkv));
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, BQ>), grid, dim3(256), lds, st, p);
        VFT_CHECK_LAUNCH("vft_attn_bwd(dkdv)");
        if (p.nsplit > 1) {
            const int64_t n = (int64_t)p.B * p.Hkv * p.Sk * (D / 4);
            hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
            VFT_CHECK_LAUNCH("vft_attn_bwd(split reduce)");
        }
    }
    return 0;
}

}  // namespace

extern "C" int vft_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, const int32_t* kv_len,
                            int B, int Hq, int Hkv, int Sq, int Sk, int D, const int64_t* q_strides,
                            const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, float scale,
                            void* stream) {
    if (check_dims(B, Hq, Hkv, Sq, Sk, D)) return 1;
    if (check_ptr("q", q) || check_ptr("k", k) || check_ptr("v", v) || check_ptr("out", out)) return 1;
    if (check_strides("q", q_strides) || check_strides("k", k_strides) || check_strides("v", v_strides) ||
        check_strides("out", o_strides))
        return 1;
    AttnParams p{};
    p.q = (const bf16_t*)q;
    p.k = (const bf16_t*)k;
    p.v = (const bf16_t*)v;
    p.o = (bf16_t*)out;
    p.lse = lse;
    p.kv_len = kv_len;
    p.B = B, p.Hq = Hq, p.Hkv = Hkv, p.Sq = Sq, p.Sk = Sk;
    p.scale = scale;
    p.sq = mk(q_strides), p.sk = mk(k_strides), p.sv = mk(v_strides), p.so = mk(o_strides);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 64: return g_attn_variant == 1 ? launch_fwd<64>(p, st) : launch_fwd3<64>(p, st);
        case 96: return g_attn_variant == 1 ? launch_fwd<96>(p, st) : launch_fwd3<96>(p, st);
        case 128: return g_attn_variant == 1 ? launch_fwd<128>(p, st) : launch_fwd2<128>(p, st);
        default: return g_attn_variant == 1 ? launch_fwd<256>(p, st) : launch_fwd2<256>(p, st);
    }
}

extern "C" int vft_attn_variant(int variant) {
    VFT_CHECK_ARG(variant >= 0 && variant <= 3, "vft_attn_variant: %d is not 0 (automatic), 1 (first-generation 4-wave kernels), 2 (8-wave dK/dV) or 3 (dq + dkdv pair where the one-pass backward would run)", variant);
    g_attn_variant = variant;
    return 0;
}

// floats the caller should allocate for vft_attn_bwd's workspace: delta, plus split-Q partials when they pay
extern "C" int64_t vft_attn_bwd_workspace(int B, int Hq, int Hkv, int Sq, int Sk, int D) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Sk <= 0) return 0;
    const int64_t delta_floats = ((int64_t)B * Hq * Sq + 3) / 4 * 4;
    const int n1 = bwd1_slices(B, Hq, Hkv, Sq, Sk, D);
    if (n1 > 0) return 2 * delta_floats + (int64_t)n1 * bwd1_plane_floats(B, Hq, Sq, D);  // -delta, -lse/scale, running-dQ plane(s)
    const int ns = dkdv_nsplit(B, Hq, Hkv, Sq, Sk, D);
    const int64_t base = delta_floats + (ns > 1 ? 2 * (int64_t)ns * B * Hkv * Sk * D : 0);
    if (stored_ds_applies(B, Hq, Sq, Sk, D))  // + dS^T (bf16) and the dQ GEMM's partial tiles
        return base + stored_ds_floats(B, Hq, Sq, Sk) + vft_wg8_batched_scratch_floats(Sk, Sq, D, B * Hq);
    return base;
}

extern "C" int vft_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                            const float* lse, void* dq, void* dk, void* dv, float* delta_ws, const int32_t* kv_len,
                            int B, int Hq, int Hkv, int Sq, int Sk, int D, const int64_t* q_strides,
                            const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                            const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                            const int64_t* dv_strides, float scale, int64_t ws_floats, void* stream) {
    if (check_dims(B, Hq, Hkv, Sq, Sk, D)) return 1;
    if (check_ptr("q", q) || check_ptr("k", k) || check_ptr("v", v) || check_ptr("out", out) ||
        check_ptr("dout", dout) || check_ptr("dq", dq) || check_ptr("dk", dk) || check_ptr("dv", dv))
        return 1;
    VFT_CHECK_ARG(lse != nullptr && delta_ws != nullptr, "vft_attn_bwd: lse / delta workspace is null");
    if (check_strides("q", q_strides) || check_strides("k", k_strides) || check_strides("v", v_strides) ||
        check_strides("out", o_strides) || check_strides("dout", do_strides) || check_strides("dq", dq_strides) ||
        check_strides("dk", dk_strides) || check_strides("dv", dv_strides))
        return 1;
    AttnParams p{};
    p.q = (const bf16_t*)q;
    p.k = (const bf16_t*)k;
    p.v = (const bf16_t*)v;
    p.o = (bf16_t*)const_cast<void*>(out);
    p.dout = (const bf16_t*)dout;
    p.dq = (bf16_t*)dq;
    p.dk = (bf16_t*)dk;
    p.dv = (bf16_t*)dv;
    p.lse = const_cast<float*>(lse);
    p.delta = delta_ws;
    p.kv_len = kv_len;
    p.B = B, p.Hq = Hq, p.Hkv = Hkv, p.Sq = Sq, p.Sk = Sk;
    p.scale = scale;
    p.sq = mk(q_strides), p.sk = mk(k_strides), p.sv = mk(v_strides), p.so = mk(o_strides);
    p.sdo = mk(do_strides), p.sdq = mk(dq_strides), p.sdk = mk(dk_strides), p.sdv = mk(dv_strides);
    // workspace = delta [B*Hq*Sq] followed (16-byte aligned) by the split-Q partials when the caller provided room
    const int64_t delta_floats = ((int64_t)B * Hq * Sq + 3) / 4 * 4;
    VFT_CHECK_ARG(ws_floats >= (int64_t)B * Hq * Sq, "vft_attn_bwd: workspace too small (%lld floats)", (long long)ws_floats);
    p.split_ws = delta_ws + delta_floats;
    hipStream_t st = (hipStream_t)stream;
    const int n1 = bwd1_slices(B, Hq, Hkv, Sq, Sk, D);
    if (n1 > 0 && ws_floats >= 2 * delta_floats + (int64_t)n1 * bwd1_plane_floats(B, Hq, Sq, D)) {
        p.nsplit = n1;
        p.nlse = delta_ws + delta_floats;
        p.split_ws = delta_ws + 2 * delta_floats;
        return launch_bwd1<64>(p, st);
    }
    p.nsplit = dkdv_nsplit(B, Hq, Hkv, Sq, Sk, D);
    if (p.nsplit > 1 && ws_floats < delta_floats + 2 * (int64_t)p.nsplit * B * Hkv * Sk * D) p.nsplit = 1;
    if (kv_len == nullptr && stored_ds_applies(B, Hq, Sq, Sk, D)) {
        const int64_t base = delta_floats + (p.nsplit > 1 ? 2 * (int64_t)p.nsplit * B * Hkv * Sk * D : 0);
        if (ws_floats >= base + stored_ds_floats(B, Hq, Sq, Sk) + vft_wg8_batched_scratch_floats(Sk, Sq, D, B * Hq)) {
            p.ds_t = reinterpret_cast<bf16_t*>(delta_ws + base);
            p.ds_ld = stored_ds_ld(Sq);
            p.dq_ws = delta_ws + base + stored_ds_floats(B, Hq, Sq, Sk);
        }
    }
    switch (D) {
        case 64: return launch_bwd<64, 64>(p, st);
        case 96: return launch_bwd<96, 64>(p, st);
        case 128: return launch_bwd<128, 64>(p, st);
        default: return launch_bwd<256, 32>(p, st);
    }
}
