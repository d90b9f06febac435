// Fused linear layer for the denoisers' projections, bf16, fp32 accumulate — gfx950.
//
//   y[M,N] = x[M,K] · W[N,K]ᵀ  (+ bias[N])  (+ ex[M,r] · ew[N,r]ᵀ)  (+ res[M,N])
//
// One launch computes what the reference's LoRALinear.forward issues as three GEMMs, a scale and an
// add (src/modules/peft/lora.py:92-104:  linear(x) + lora_up(lora_down(x)) * (alpha/rank)):
// ex = lora_down(x) (rank-r activations, vft_lora_down) and ew = (alpha/rank)·lora_up enter the same
// accumulators as r extra K columns, so the adapter costs one MFMA step instead of a read-modify-write
// pass over y.  The optional residual is the `x + attn(x)` / `x + ff(x)` add of the transformer blocks
// (src/models/sdxl/denoiser.py:273-284), applied to the rounded output exactly as the reference does.
// The backward data gradient dx = dy·W + (dy·B)·(sA) is the same kernel on the transposed frozen weight.
//
// Structure (CDNA4): 256(M) x 256(N) or 256 x 128 output tile, 512 threads = 8 waves, each wave 128x64 or
// 64x64 of v_mfma_f32_32x32x16_bf16 with the weight as the MFMA A operand (an accumulator register group =
// 4 consecutive output features of one token).  K advances 32 at a time through a ring of LDS stages filled
// by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip); the two 4-wave groups run the same program one
// phase apart (see the schedule comment at the kernel), so each SIMD always has one wave in its MFMA phase.
// The LDS image is lane-linear (a DMA writes base + lane*16), so the bank swizzle is applied to the per-lane
// SOURCE chunk and again on the fragment read: chunk' = chunk ^ ((row >> 2) & 3), conflict-free for
// ds_read_b128 with row = lane & 31 on 64-byte rows.  The epilogue transposes the tile through LDS so that y
// is written in full row segments.  Workgroup ids are remapped so that the workgroups resident on one XCD
// share weight tiles (and activation rows) through that XCD's L2.
//
// Measured (MI355X, bf16, random data): 730-1000 TFLOP/s on the SDXL / AuraFlow projection shapes, 70-75 %
// of hipBLASLt's best solution; the rank extension is free.  The denoisers therefore still route plain
// projections to the tuned library (vision_ft_amd/tuning.py) — this kernel is the base for the fused
// epilogues (LoRA, residual, NF4) once its main loop closes the remaining gap.
#include <cstdlib>

#include "vft_common.h"
#include "linear8.h"

namespace {

constexpr int NTHREADS = 512;

// kernel choice override for tuning / A-B runs (vft_linear_variant): 0 = automatic, -1 = first-generation kernel,
// 1..3 = second-generation kernel (linear8.hip) with that tile configuration
int g_variant = 0;


struct LinParams {
    const bf16_t* x;     // [M][ldx]
    const bf16_t* w;     // [N][ldw]
    bf16_t* y;           // [M][ldy]
    const bf16_t* bias;  // [N] or null
    const bf16_t* res;   // [M][ldres] or null
    const bf16_t* ex;    // [M][r] or null
    const bf16_t* ew;    // [N][r]
    int M, N, K, r;
    int64_t ldx, ldw, ldy, ldres;
    // NF4 weight modes: logical weight [QN][QK] row-major, bitsandbytes packing (element 2i in the high nibble of
    // byte i), one fp32 absmax per 64 consecutive elements (double-quantised absmax already expanded by the host)
    const uint8_t* q_packed;
    const float* q_absmax;
    const float* q_code;  // [16]
    int QN, QK;
    int cH, cW, cC;      // CONV mode: image height / width / input channels (x is [B*H*W][ldx], K = 9 * cC)
};

// source of the zero padding in CONV mode (an LDS-DMA cannot skip lanes: masked lanes would leave stale data)
__device__ __attribute__((aligned(16))) bf16_t g_zero_page[32];

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// LDS-DMA through INLINE ASM (global_load_lds_dwordx4, M0 = wave-uniform LDS destination, saved / restored in the
// statement).  With the builtin the compiler sees a pending LDS write on the VM counter and drains it (`s_waitcnt vmcnt(0)`)
// in front of the next LDS access it cannot disambiguate — in the NF4 modes that is the dequantisation's look-up-table read
// right behind the counted staging wait (seen in the round-2 ISA: `vmcnt(4)` followed by `vmcnt(0)` in every stage), in the
// transposed mode every ds_read_b64_tr_b16.  Hidden in asm, the ring's counted waits below are the only ones.
__device__ __forceinline__ void glds16(const bf16_t* g, bf16_t* l) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)l);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst));
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// rank-r operand fragment: 8 consecutive k of row `row` starting at k (zero beyond r); r in {4, 8, 16, 32}
__device__ __forceinline__ bf16x8 ext_frag(const bf16_t* base, int64_t row, int k, int r) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
    if (k < r) {
        const bf16_t* p = base + row * r + k;
        if (r - k >= 8) {
            v = *reinterpret_cast<const bf16x8*>(p);
        } else {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = t[i];
        }
    }
    return v;
}

// NF4 staging loads are issued as inline asm so that hipcc neither tracks them (it would drain the whole DMA
// pipeline with vmcnt(0) at their first use) nor moves their uses above our counted wait: the wait takes the
// destination registers as in/out operands.
__device__ __forceinline__ void asm_load_b64(u32x2& d, const void* p) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load_b128(u32x4& d, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load_b32(float& d, const void* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}

// MFMA 32x32x16 A-operand fragment (32 rows n x 16 k) out of a [k][n] row-major LDS tile by two transposed reads
// (ds_read_b64_tr_b16): lane half h gets k = 8h .. 8h+7 of row n = lane & 31 — the same k order as a plain
// ds_read_b128 of a [n][k] tile, so the activation operand needs no permutation.
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
__device__ __forceinline__ int tr_lane_off_nat(int ld, int lane) {
    const int h = lane >> 5, i = lane & 15;
    return (8 * h + (i >> 2)) * ld + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
}
__device__ __forceinline__ bf16x8 tr_frag_nat(const bf16_t* lane_base, int ld, int row0, int col0) {
    const bf16_t* a0 = lane_base + row0 * ld + col0;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a0 + 4 * ld));
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile configuration: BM x BN output tile, 8 waves as WAVES_M x WAVES_N.  K advances in steps of
// KST (32 or 64) per LDS stage, consumed in 32-deep phases (two MFMA k-steps); a ring of NSTAGE stages.
//   <256, 256, 2>  wave tile 128(M) x 64(N): 12 fragment reads, 4 DMAs, 16 MFMAs per K-step   (wide N)
//   <256, 128, 4>  wave tile  64(M) x 64(N):  8 fragment reads, 3 DMAs,  8 MFMAs per K-step   (narrow N)
template <int BM_, int BN_, int WAVES_M_, int NSTAGE_, int WAVES_PER_SIMD_, bool DMA_IN_M_, bool CONV_ = false, int KST_ = 32,
          int WMODE_ = 0>
struct Cfg {
    static constexpr bool CONV = CONV_;
    static constexpr int WMODE = WMODE_;  // 0: bf16 weight by LDS-DMA; 1: NF4 as stored; 2: NF4 transposed (data gradient)
    static constexpr int KST = KST_;                   // K per LDS stage: 32 (64-byte rows) or 64 (full 128-byte lines)
    static constexpr int RPD = 512 / KST_, LPR = KST_ / 8;  // rows per DMA instruction, lanes per row
    static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = 8 / WAVES_M_, NSTAGE = NSTAGE_;
    static constexpr int WAVES_PER_SIMD = WAVES_PER_SIMD_;
    static constexpr bool DMA_IN_M = DMA_IN_M_;
    static constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, FM = TM / 32, FN = TN / 32;
    static constexpr int WT_LD = BN + 8;  // WMODE 2: the weight tile is kept in W's own orientation, [64 k'][BN n' + pad]
    static constexpr int A_STAGE = BM * KST, B_STAGE = WMODE_ == 2 ? KST * WT_LD : BN * KST, STAGE = A_STAGE + B_STAGE;  // elements
    static constexpr int DMA_A = BM / (8 * RPD), DMA_B = WMODE_ ? 0 : BN / (8 * RPD);
    static constexpr int QW = BN / 64;                       // NF4: packed dwords per thread per stage (8 weights each)
    static constexpr int DMA = DMA_A + (WMODE_ ? 2 : DMA_B);  // VM operations per wave per stage (NF4: codes + absmax loads)
    static constexpr int C_LD = BN + 8, EP_ROWS = 128;                                     // epilogue image
    static constexpr size_t LDS_BYTES = (size_t)NSTAGE * STAGE * sizeof(bf16_t);
    static_assert((size_t)EP_ROWS * C_LD * sizeof(bf16_t) <= LDS_BYTES, "epilogue image must fit in the stage buffers");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_barrier" ::: "memory");
}

// Main-loop schedule.  The 8 waves form two groups of four (one wave of each group per SIMD) that run
// the same program one phase apart:
//     group 0:      R0 | M0 | R1 | M1 | ...            R = fragment ds_reads + LDS-DMA issue for step t+3
//     group 1:  .   .  | R0 | M0 | R1 | ...            M = the step's MFMAs;   `|` = s_barrier (all 8 waves)
// so on every SIMD one wave is issuing MFMAs while the other fills its fragments — the matrix pipe never
// waits for LDS.  Hazards: R ends with lgkmcnt(0) (its reads are done before the next rendezvous, so the
// stage may be re-filled one phase later) and a counted vmcnt that retires this wave's DMAs of step t+1
// two rendezvous before any wave reads them; the DMAs of steps t+2, t+3 stay in flight across barriers.
template <class C>
__global__ __launch_bounds__(NTHREADS, C::WAVES_PER_SIMD) void linear_kernel(LinParams p) {
    constexpr int BM = C::BM, BN = C::BN, FM = C::FM, FN = C::FN, STAGE = C::STAGE, A_STAGE = C::A_STAGE;
    constexpr int KS = C::KST, RPD = C::RPD, LPR = C::LPR;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, h = lane >> 5;
    // waves 0-3 / 4-7 are the two phase groups; inside a group 4 waves tile WAVES_M x WAVES_N / 2 ... the
    // (wm, wn) assignment only needs to be a bijection of 0..7
    const int wm = wv / C::WAVES_N, wn = wv % C::WAVES_N;
    const int grp = wv >> 2;

    // ---- workgroup -> tile: XCD-contiguous ids, then bands of <= 8 N-tiles walked M-major ----------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    {
        const int nwg = tiles_m * tiles_n, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
        const int lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
        const int GN = tiles_n < 8 ? tiles_n : 8;
        const int band = lin / (GN * tiles_m);
        const int nb = min(GN, tiles_n - band * GN);
        const int within = lin - band * GN * tiles_m;
        tn = band * GN + within % nb;
        tm = within / nb;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA source pointers: one DMA = 16 rows x 64 B, lane -> (row = lane>>2, slot = lane&3);
    //      the image is lane-linear, the bank swizzle sits on the source chunk: chunk = slot ^ ((row>>2)&3)
    const bf16_t* ga[C::DMA_A];
    const bf16_t* gb[C::DMA_B ? C::DMA_B : 1];
    const bf16_t* gz[C::DMA_A];  // CONV only
    unsigned vmask[C::DMA_A];    // CONV only
    {
        const int lrow = lane / LPR, slot = lane % LPR;
#pragma unroll
        for (int q = 0; q < C::DMA_A; ++q) {
            const int row = (wv * C::DMA_A + q) * RPD + lrow;
            const int ch = slot ^ (KS == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7));
            const int pm = min(m0 + row, p.M - 1);
            ga[q] = p.x + (int64_t)pm * p.ldx + ch * 8;
            if (C::CONV) {  // which of the 9 taps fall inside the image for this output pixel
                const int rem = pm % (p.cH * p.cW), oh = rem / p.cW, ow = rem - oh * p.cW;
                unsigned m = 0;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ih = oh + tap / 3 - 1, iw = ow + tap % 3 - 1;
                    if (ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW) m |= 1u << tap;
                }
                vmask[q] = m;
                gz[q] = g_zero_page + ch * 8;
            }
        }
#pragma unroll
        for (int q = 0; q < C::DMA_B; ++q) {
            const int row = (wv * C::DMA_B + q) * RPD + lrow;
            const int ch = slot ^ (KS == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7));
            gb[q] = p.w + (int64_t)min(n0 + row, p.N - 1) * p.ldw + ch * 8;
        }
    }
    // ---- NF4 weight staging (WMODE 1 / 2): packed codes + absmax -> registers -> dequantised bf16 tile in LDS ---
    constexpr int QW = C::QW;            // packed dwords (8 weights each) per thread per 64-deep stage
    __shared__ float2 code2_s[256];  // byte -> (code[hi nibble], code[lo nibble]): one LDS lookup per two weights
    typedef __attribute__((ext_vector_type(QW))) unsigned int qvec_t;
    qvec_t qreg;
    float qam = 0.f;
    const uint8_t* qcp = nullptr;  // this thread's packed codes at k0 = 0
    const float* qap = nullptr;    // this thread's absmax at k0 = 0
    int64_t qcs = 0, qas = 0;      // byte / element advance per stage
    int qrow = 0, qseg = 0;
    if (C::WMODE != 0) {
        if (tid < 256) code2_s[tid] = make_float2(p.q_code[tid >> 4], p.q_code[tid & 15]);
        if (C::WMODE == 1) {  // tile rows = weight rows n, 64 k per stage = one quantisation block per row
            constexpr int TPR = 8 / QW;  // threads per row
            qrow = tid / TPR, qseg = tid % TPR;
            const int64_t e0 = (int64_t)min(n0 + qrow, p.N - 1) * p.QK + qseg * 8 * QW;
            qcp = p.q_packed + (e0 >> 1);
            qap = p.q_absmax + (e0 >> 6);
            qcs = KS / 2, qas = 1;
        } else {              // tile rows = weight columns n', stage = 64 weight rows k'
            qrow = tid >> 3, qseg = tid & 7;
            const int gcol = min(n0 + qseg * 8 * QW, p.QK - 8 * QW);
            const int64_t e0 = (int64_t)qrow * p.QK + gcol;
            qcp = p.q_packed + (e0 >> 1);
            qap = p.q_absmax + (e0 >> 6);
            qcs = (int64_t)KS * p.QK / 2, qas = (int64_t)KS * p.QK / 64;
        }
    }
    auto nf4_store = [&](int slot) {
        bf16_t* sb = smem + slot * STAGE + A_STAGE;
#pragma unroll
        for (int w = 0; w < QW; ++w) {
            const unsigned int d = qreg[w];
            bf16_t v[8];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float2 c2 = code2_s[(d >> (8 * bb)) & 0xffu];
                v[2 * bb] = (bf16_t)(c2.x * qam);
                v[2 * bb + 1] = (bf16_t)(c2.y * qam);
            }
            if (C::WMODE == 1) {
                const int chunk = qseg * QW + w;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[e];
                *reinterpret_cast<bf16x8*>(sb + qrow * KS + ((chunk ^ ((qrow >> 1) & 7)) << 3)) = o;
            } else {  // W's own orientation: row k' = qrow, columns n' = qseg*8*QW + 8w .. +7; read back transposed
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[e];
                *reinterpret_cast<bf16x8*>(sb + qrow * C::WT_LD + qseg * 8 * QW + w * 8) = o;
            }
        }
    };
    auto wait_q = [&]() {  // the staging loads have landed (everything issued before them too)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(qreg), "+v"(qam) : "n"(C::DMA_A) : "memory");
    };

    constexpr int NSTAGE = C::NSTAGE;
    // CONV: K index = tap * Cin + c (the channels-last weight layout); K-steps are issued in increasing order,
    // so the (tap, channel offset) of the next step is tracked incrementally
    int ctap = 0, cc0 = 0;
    // NF4: this thread's packed codes + absmax of stage t -> registers.  Program order per stage is
    // [staging loads][the stage's A DMAs], so `wait_q` (vmcnt(DMA_A)) retires exactly the staging loads.
    auto issue_q = [&](int t) {
        if (QW == 2)
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(qreg) : "v"(qcp + (int64_t)t * qcs) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qreg) : "v"(qcp + (int64_t)t * qcs) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(qam) : "v"(qap + (int64_t)t * qas) : "memory");
    };
    auto issue = [&](int t, int slot) {
        bf16_t* st = smem + slot * STAGE;
        const int k0 = t * KS;
        if (C::CONV) {
            const int64_t off = (int64_t)((ctap / 3 - 1) * p.cW + (ctap % 3 - 1)) * p.ldx + cc0;
#pragma unroll
            for (int q = 0; q < C::DMA_A; ++q)
                glds16(((vmask[q] >> ctap) & 1u) ? ga[q] + off : gz[q], st + (wv * C::DMA_A + q) * RPD * KS);
            cc0 += KS;
            if (cc0 == p.cC) {
                cc0 = 0;
                ++ctap;
            }
        } else {
#pragma unroll
            for (int q = 0; q < C::DMA_A; ++q) glds16(ga[q] + k0, st + (wv * C::DMA_A + q) * RPD * KS);
        }
#pragma unroll
        for (int q = 0; q < C::DMA_B; ++q) glds16(gb[q] + k0, st + A_STAGE + (wv * C::DMA_B + q) * RPD * KS);
    };

    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets (elements): row * KS + ((chunk ^ sw) << 3); chunk = 2 * kk + h for MFMA k-step kk
    const int sw = KS == 32 ? ((r32 >> 2) & 3) : ((r32 >> 1) & 7);
    const int xrow = (wm * C::TM + r32) * KS, wrow = A_STAGE + (wn * C::TN + r32) * KS;
    const int wtr = tr_lane_off_nat(C::WT_LD, lane);  // WMODE 2

    const int nkt = p.K / KS;
    // prologue: stages 0 .. NSTAGE-2 in flight, stage 0 landed for everyone, then stagger the groups
    if (C::WMODE != 0) __syncthreads();  // code2_s
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nkt) {
            if (C::WMODE != 0) issue_q(s0);
            issue(s0, s0);
            if (C::WMODE != 0) {
                wait_q();
                nf4_store(s0);
            }
        }
    if (C::WMODE != 0) {
        // in-order retirement: only the last issued stage's DMAs can still be outstanding
        if (NSTAGE > 2 && nkt >= NSTAGE - 1) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::DMA_A) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        if (NSTAGE - 1 < nkt) issue_q(NSTAGE - 1);  // staging loads run one phase ahead of their stage's DMAs
    } else if (nkt >= NSTAGE - 1) {
        wait_vm<(NSTAGE - 2) * C::DMA>();
    } else {
        wait_vm<0>();
    }
    raw_barrier();
    if (grp == 1) raw_barrier();

    // vmcnt that retires this wave's DMAs of stage t+1 in the last phase of stage t: stages up to t+NSTAGE-1
    // (issued in R) or t+NSTAGE-2 (issued in M) are outstanding behind it
    constexpr int INFLIGHT = C::WMODE != 0 ? (NSTAGE > 2 ? C::DMA_A + 2 : 2)  // (the staging-load wait already retired older
                                                                             // stages; 2 = the next stage's staging loads)
                                           : (C::DMA_IN_M ? NSTAGE - 3 : NSTAGE - 2) * C::DMA;
    constexpr int PH = KS / 32;  // phases (32-deep) per stage
    int rs = 0;            // ring slot of stage t
    int ws = NSTAGE - 1;   // ring slot of stage t + NSTAGE - 1
    for (int t = 0; t < nkt; ++t) {
        const bf16_t* st = smem + rs * STAGE;
        const bool more = t + NSTAGE - 1 < nkt;
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            // ---- R ----
            const int co0 = ((4 * ph + h) ^ sw) << 3, co1 = ((4 * ph + 2 + h) ^ sw) << 3;
            bf16x8 fw[2][FN], fx[2][FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                if (C::WMODE == 2) {
                    fw[0][i] = tr_frag_nat(st + A_STAGE + wtr, C::WT_LD, 32 * ph, wn * C::TN + i * 32);
                    fw[1][i] = tr_frag_nat(st + A_STAGE + wtr, C::WT_LD, 32 * ph + 16, wn * C::TN + i * 32);
                } else {
                    fw[0][i] = *reinterpret_cast<const bf16x8*>(st + wrow + i * 32 * KS + co0);
                    fw[1][i] = *reinterpret_cast<const bf16x8*>(st + wrow + i * 32 * KS + co1);
                }
            }
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                fx[0][j] = *reinterpret_cast<const bf16x8*>(st + xrow + j * 32 * KS + co0);
                fx[1][j] = *reinterpret_cast<const bf16x8*>(st + xrow + j * 32 * KS + co1);
            }
            if (ph == 0 && !C::DMA_IN_M && more) issue(t + NSTAGE - 1, ws);
            if (C::WMODE != 0 && ph == 1 && more) {  // two phases after their issue: dequantise into the stage
                wait_q();
                nf4_store(ws);
                if (t + NSTAGE < nkt) issue_q(t + NSTAGE);
            }
            if (ph == PH - 1) {
                if (more && C::WMODE != 0 && !(t + NSTAGE < nkt)) {  // no staging loads were issued behind the DMAs
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTAGE > 2 ? C::DMA_A : 0) : "memory");
                } else if (more) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            raw_barrier();
            // ---- M ----
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
                        acc[i][j] = mfma32(fw[ks][i], fx[ks][j], acc[i][j]);
                        if (C::DMA_IN_M && ph == 0 && ks == 0 && i == 0 && j == 0 && more) issue(t + NSTAGE - 1, ws);
                    }
            __builtin_amdgcn_s_setprio(0);
            raw_barrier();
        }
        rs = rs + 1 == NSTAGE ? 0 : rs + 1;
        ws = ws + 1 == NSTAGE ? 0 : ws + 1;
    }
    if (grp == 0) raw_barrier();

    // ---- rank-r extension: r extra K columns straight from global into fragments ---------------------
    if (p.ex != nullptr) {
        for (int k0 = 0; k0 < p.r; k0 += 16) {
            bf16x8 fw[FN], fx[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[i] = ext_frag(p.ew, min(n0 + wn * C::TN + i * 32 + r32, p.N - 1), k0 + 8 * h, p.r);
#pragma unroll
            for (int j = 0; j < FM; ++j) fx[j] = ext_frag(p.ex, min(m0 + wm * C::TM + j * 32 + r32, p.M - 1), k0 + 8 * h, p.r);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
        }
    }

    // ---- epilogue: accumulators (+bias) -> bf16 image in LDS (128 token rows per pass) -> 16-byte
    //      row-contiguous stores --------------------------------------------------------------------
    bf16_t* cs = smem;
    constexpr int C_LD = C::C_LD, CH = BN / 8;  // 16-byte chunks per image row
#pragma unroll
    for (int pass = 0; pass < BM / C::EP_ROWS; ++pass) {
        __syncthreads();
        if ((wm * C::TM) / C::EP_ROWS == pass) {
            const int trow = wm * C::TM - pass * C::EP_ROWS;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int fl = wn * C::TN + i * 32 + 8 * g + 4 * h;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias != nullptr && n0 + fl < p.N) {
                        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p.bias + n0 + fl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = (float)t[e];
                    }
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
                        bf16x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc[i][j][4 * g + e] + bv[e]);
                        *reinterpret_cast<bf16x4*>(cs + (trow + j * 32 + r32) * C_LD + fl) = v;
                    }
                }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (C::EP_ROWS * CH) / NTHREADS; ++q) {
            const int idx = tid + NTHREADS * q;
            const int row = idx / CH, ch = idx % CH;
            const int m = m0 + pass * C::EP_ROWS + row, n = n0 + ch * 8;
            if (m < p.M && n < p.N) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(cs + row * C_LD + ch * 8);
                if (p.res != nullptr) {
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(p.res + (int64_t)m * p.ldres + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)rv[e]);
                }
                *reinterpret_cast<bf16x8*>(p.y + (int64_t)m * p.ldy + n) = v;
            }
        }
    }
}

template <class C>
void launch_linear(const LinParams& p, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(linear_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::LDS_BYTES);
        attr = true;
    }
    const unsigned grid = (unsigned)(((p.M + C::BM - 1) / C::BM) * ((p.N + C::BN - 1) / C::BN));
    hipLaunchKernelGGL(linear_kernel<C>, dim3(grid), dim3(NTHREADS), C::LDS_BYTES, st, p);
}

}  // namespace

// Measured tile choices per problem (vft_linear_tile_hint): consulted before the analytic model of vft_lin8_pick.
// Key = (M, N, K, conv); a handful to a few hundred entries, set once at start-up from the committed table
// (vision_ft_amd/tuning_tables/gfx950_lin8_tiles.csv); read on every launch (host side only).
#include <mutex>
#include <unordered_map>
namespace {
std::unordered_map<uint64_t, int> g_tile_hints;
std::mutex g_tile_mu;
inline uint64_t tile_key(int M, int N, int K, int conv) {
    // M < 2^31, N, K < 2^20 for every operand this library accepts (2 GiB descriptors); mixed into 64 bits
    uint64_t h = (uint64_t)(uint32_t)M * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)N << 21) ^ ((uint64_t)(uint32_t)K << 42) ^ (uint64_t)(conv & 1);
    return h * 0xD6E8FEB86659FD93ull + (uint64_t)(uint32_t)N * 31u + (uint64_t)(uint32_t)K;
}
int pick_tile(int M, int N, int K, int conv) {
    {
        std::lock_guard<std::mutex> g(g_tile_mu);
        auto it = g_tile_hints.find(tile_key(M, N, K, conv));
        if (it != g_tile_hints.end()) return it->second;
    }
    return vft_lin8_pick(M, N, K);
}
}  // namespace

// cfg 1..3: one launch on that tile; 11..13: whole rounds on tile cfg - 10, the remaining rows as a second launch
static void launch_lin8_cfg(Lin8Params q, int cfg, int conv, hipStream_t st) {
    if (cfg >= 11) {
        const int main_cfg = cfg - 10;
        const int M1 = vft_lin8_split_rows(q.M, q.N, main_cfg);
        if (M1 > 0) {
            Lin8Params a = q, b = q;
            a.M = M1;
            a.x_rows = b.x_rows = q.x_rows > 0 ? q.x_rows : q.M;
            b.m_begin = M1;
            b.q_packed = nullptr;  // NF4: the first launch has dequantised the weight into the scratch; the tail reads it as bf16
            vft_lin8_launch(a, main_cfg, conv, st);
            vft_lin8_launch(b, vft_lin8_pick_plain(q.M - M1, q.N, q.K), conv, st);
            return;
        }
        cfg = main_cfg;
    }
    vft_lin8_launch(q, cfg, conv, st);
}

extern "C" int vft_linear_tile_hint(int M, int N, int K, int conv, int cfg) {
    VFT_CHECK_ARG(M > 0 && N > 0 && K > 0 && ((cfg >= 0 && cfg <= 3) || (cfg >= 11 && cfg <= 13)),
                  "vft_linear_tile_hint: need M,N,K>0 and cfg in 0..3 or 11..13 (0 clears)");
    std::lock_guard<std::mutex> g(g_tile_mu);
    if (cfg == 0)
        g_tile_hints.erase(tile_key(M, N, K, conv));
    else
        g_tile_hints[tile_key(M, N, K, conv)] = cfg;
    return 0;
}

extern "C" int vft_linear_variant(int variant) {
    VFT_CHECK_ARG((variant >= -1 && variant <= 3) || (variant >= 11 && variant <= 13), "vft_linear_variant: %d is not in -1..3 / 11..13", variant);
    g_variant = variant;
    return 0;
}

extern "C" int vft_linear_fwd(const void* x, const void* w, void* y, const void* bias, const void* residual,
                              const void* ext_x, const void* ext_w, int M, int N, int K, int r, int64_t ldx, int64_t ldw,
                              int64_t ldy, int64_t ldres, void* stream) {
    VFT_CHECK_ARG(x && w && y, "vft_linear_fwd: null pointer");
    VFT_CHECK_ARG(M > 0 && N > 0 && K > 0, "vft_linear_fwd: non-positive dimension");
    VFT_CHECK_ARG(K % 32 == 0, "vft_linear_fwd: K=%d must be a multiple of 32", K);
    VFT_CHECK_ARG(N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0, "vft_linear_fwd: N, ldx, ldw, ldy must be multiples of 8");
    VFT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "vft_linear_fwd: pointers must be 16-byte aligned");
    VFT_CHECK_ARG(bias == nullptr || ((uintptr_t)bias & 7) == 0, "vft_linear_fwd: misaligned bias");
    VFT_CHECK_ARG(residual == nullptr || (ldres % 8 == 0 && ((uintptr_t)residual & 15) == 0), "vft_linear_fwd: bad residual pointer / stride");
    VFT_CHECK_ARG((ext_x == nullptr) == (ext_w == nullptr), "vft_linear_fwd: ext_x and ext_w go together");
    VFT_CHECK_ARG(ext_x == nullptr || ((r == 4 || r == 8 || r == 16 || r == 32) && (((uintptr_t)ext_x | (uintptr_t)ext_w) & 15) == 0),
                  "vft_linear_fwd: rank extension needs r in {4, 8, 16, 32} and 16-byte aligned operands (r=%d)", r);
    LinParams p{};
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w, p.y = (bf16_t*)y, p.bias = (const bf16_t*)bias;
    p.res = (const bf16_t*)residual, p.ex = (const bf16_t*)ext_x, p.ew = (const bf16_t*)ext_w;
    p.M = M, p.N = N, p.K = K, p.r = ext_x ? r : 0, p.ldx = ldx, p.ldw = ldw, p.ldy = ldy, p.ldres = ldres;
    Lin8Params q{};
    q.x = p.x, q.w = p.w, q.y = p.y, q.bias = p.bias, q.res = p.res, q.ex = p.ex, q.ew = p.ew;
    q.M = M, q.N = N, q.K = K, q.r = p.r, q.ldx = ldx, q.ldw = ldw, q.ldy = ldy, q.ldres = ldres;
    if (g_variant >= 0 && q.r <= 16 && vft_lin8_fits(q, false)) {
        launch_lin8_cfg(q, g_variant > 0 ? g_variant : pick_tile(M, N, K, 0), false, (hipStream_t)stream);
        VFT_CHECK_LAUNCH("vft_linear_fwd");
        return 0;
    }
    // wide tile when N fills it (>= 75 % of the last 256-wide tile column is real work)
    // 64-deep stages (full 128-byte lines per DMA row) whenever K allows: 5-15 % faster than 32-deep ones
    const int rem = N % 256;
    const bool wide = N >= 1024 && (rem == 0 || rem > 192);
    if (K % 64 == 0) {
        if (wide)
            launch_linear<Cfg<256, 256, 2, 2, 2, false, false, 64>>(p, (hipStream_t)stream);
        else
            launch_linear<Cfg<256, 128, 4, 3, 2, false, false, 64>>(p, (hipStream_t)stream);
    } else if (wide)
        launch_linear<Cfg<256, 256, 2, 4, 2, true>>(p, (hipStream_t)stream);
    else
        launch_linear<Cfg<256, 128, 4, 3, 4, false>>(p, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_linear_fwd");
    return 0;
}

// 3x3 convolution, stride 1, padding 1, channels-last, as an implicit GEMM on the kernel above:
//   y[b,oh,ow,:] = bias + Σ_{kh,kw} x[b,oh+kh-1,ow+kw-1,:] · w[:,kh,kw,:]ᵀ
// M = B*H*W output pixels, N = Cout, K = 9*Cin with k = (kh*3+kw)*Cin + c, i.e. the weight exactly as a
// channels-last nn.Conv2d stores it ([Cout][3][3][Cin]).  The activation operand is gathered by the LDS-DMA
// source addresses (one shifted pixel row per tap; out-of-image taps read a zero page).  Replaces the cuDNN /
// MIOpen convolutions of the SDXL ResidualBlock and Upsample (src/models/sdxl/denoiser.py:466-551); the data
// gradient is the same call on the flipped, transposed weight.
extern "C" int vft_conv3x3_fwd(const void* x, const void* w, void* y, const void* bias, int B, int H, int W, int Cin,
                               int Cout, void* stream) {
    VFT_CHECK_ARG(x && w && y, "vft_conv3x3_fwd: null pointer");
    VFT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "vft_conv3x3_fwd: non-positive dimension");
    VFT_CHECK_ARG(Cin % 32 == 0 && Cout % 8 == 0, "vft_conv3x3_fwd: needs Cin %% 32 == 0 and Cout %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    VFT_CHECK_ARG((int64_t)B * H * W < (1LL << 31), "vft_conv3x3_fwd: too many pixels");
    VFT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "vft_conv3x3_fwd: pointers must be 16-byte aligned");
    VFT_CHECK_ARG(bias == nullptr || ((uintptr_t)bias & 7) == 0, "vft_conv3x3_fwd: misaligned bias");
    LinParams p{};
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w, p.y = (bf16_t*)y, p.bias = (const bf16_t*)bias;
    p.M = B * H * W, p.N = Cout, p.K = 9 * Cin, p.r = 0, p.ldx = Cin, p.ldw = 9 * (int64_t)Cin, p.ldy = Cout, p.ldres = 0;
    p.cH = H, p.cW = W, p.cC = Cin;
    Lin8Params q{};
    q.x = p.x, q.w = p.w, q.y = p.y, q.bias = p.bias;
    q.M = p.M, q.N = p.N, q.K = p.K, q.r = 0, q.ldx = p.ldx, q.ldw = p.ldw, q.ldy = p.ldy, q.ldres = 0;
    q.cH = H, q.cW = W, q.cC = Cin;
    if (Cin % 64 == 0 && g_variant >= 0 && vft_lin8_fits(q, true)) {
        launch_lin8_cfg(q, g_variant > 0 ? g_variant : pick_tile(p.M, p.N, p.K, 1), true, (hipStream_t)stream);
        VFT_CHECK_LAUNCH("vft_conv3x3_fwd");
        return 0;
    }
    const int rem = Cout % 256;
    const bool wide = Cout >= 1024 && (rem == 0 || rem > 192);
    if (Cin % 64 == 0 && Cout >= 640) {  // (the 320-channel layers measure the same or slightly better 32-deep)
        if (wide)
            launch_linear<Cfg<256, 256, 2, 2, 2, false, true, 64>>(p, (hipStream_t)stream);
        else
            launch_linear<Cfg<256, 128, 4, 3, 2, false, true, 64>>(p, (hipStream_t)stream);
    } else if (wide)
        launch_linear<Cfg<256, 256, 2, 4, 2, true, true>>(p, (hipStream_t)stream);
    else
        launch_linear<Cfg<256, 128, 4, 3, 4, false, true>>(p, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_conv3x3_fwd");
    return 0;
}

// 3x3 convolution, STRIDE 2, padding 1, channels-last (the SDXL Downsample, src/models/sdxl/denoiser.py:371-408):
//   y[b,oh,ow,:] = bias + Σ_{kh,kw} x[b,2oh+kh-1,2ow+kw-1,:] · w[:,kh,kw,:]ᵀ      H, W even; output H/2 x W/2
// The same implicit GEMM as vft_conv3x3_fwd — M = B*(H/2)*(W/2) output pixels, the activation rows gathered from the
// H x W input by the LDS-DMA source addresses — so the two stride-2 convolutions of the UNet leave MIOpen (whose kernels
// for them accumulate with atomics: the last non-reproducible launches of the SDXL step).  Their data / weight gradients
// run on vft_conv3x3_fwd / vft_conv3x3_wgrad over the zero-upsampled output gradient (kernels.py::_Conv3x3S2).
extern "C" int vft_conv3x3_s2_fwd(const void* x, const void* w, void* y, const void* bias, int B, int H, int W, int Cin,
                                  int Cout, void* stream) {
    VFT_CHECK_ARG(x && w && y, "vft_conv3x3_s2_fwd: null pointer");
    VFT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "vft_conv3x3_s2_fwd: non-positive dimension");
    VFT_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "vft_conv3x3_s2_fwd: H=%d and W=%d must be even", H, W);
    VFT_CHECK_ARG(Cin % 64 == 0 && Cout % 8 == 0, "vft_conv3x3_s2_fwd: needs Cin %% 64 == 0 and Cout %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    VFT_CHECK_ARG((int64_t)B * H * W < (1LL << 31), "vft_conv3x3_s2_fwd: too many pixels");
    VFT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "vft_conv3x3_s2_fwd: pointers must be 16-byte aligned");
    VFT_CHECK_ARG(bias == nullptr || ((uintptr_t)bias & 7) == 0, "vft_conv3x3_s2_fwd: misaligned bias");
    Lin8Params q{};
    q.x = (const bf16_t*)x, q.w = (const bf16_t*)w, q.y = (bf16_t*)y, q.bias = (const bf16_t*)bias;
    q.M = B * (H / 2) * (W / 2), q.N = Cout, q.K = 9 * Cin, q.r = 0, q.ldx = Cin, q.ldw = 9 * (int64_t)Cin, q.ldy = Cout, q.ldres = 0;
    q.cH = H, q.cW = W, q.cC = Cin, q.cHo = H / 2, q.cWo = W / 2, q.x_rows = B * H * W;
    VFT_CHECK_ARG(vft_lin8_fits(q, 2), "vft_conv3x3_s2_fwd: operands beyond the 2 GiB the buffer descriptors address");
    // (tile: the plain one-launch choice; cfg 1 maps to the 256x128 instantiation inside vft_lin8_launch)
    vft_lin8_launch(q, g_variant > 0 && g_variant <= 3 ? g_variant : vft_lin8_pick_plain(q.M, q.N, q.K), 2, (hipStream_t)stream);
    VFT_CHECK_LAUNCH("vft_conv3x3_s2_fwd");
    return 0;
}

// NF4-weight linear on the same kernel: the packed 4-bit codes and one fp32 absmax per 64 weights are what travels
// from HBM; the bf16 weight exists only as LDS tiles (dequantised by the staging threads one phase before use).
//   transpose_w = 0:  y[M, out] = x[M, in]  . W[out, in]^T (+bias) (+ ext_x . ext_w^T) (+residual)
//   transpose_w = 1:  y[M, in]  = x[M, out] . W[out, in]   (+ ext ...)                       (data gradient)
// Replaces bnb.matmul_4bit of BnbLinear4bit (src/modules/quant/bnb.py:37-129) and, with the rank extension, the
// whole NF4-base + LoRA forward of a QLoRA layer in one launch.  absmax: already expanded to fp32 by the caller
// (double-quantised checkpoints: code2[q8] * absmax2[blk / 256] + offset, once at load time).
// Round 4: with a workspace (vft_linear_nf4_workspace bytes, its FIRST 16 KiB zero before the first use, re-usable by
// every NF4 layer of a stream) the launch dequantises the weight ONCE into it — all workgroups share the work, meet at a
// grid barrier and run the bf16 main loop of linear8.hip on the result — instead of once per row tile inside the K-loop
// (the form without a workspace, kept for operands the second-generation kernel does not take).
extern "C" int64_t vft_linear_nf4_workspace(int out_features, int in_features) {
    return (((int64_t)out_features * in_features * 2 + 255) & ~255ll) + 16384;  // 16 KiB header: barrier words (+ cycle stamps of the timing experiments)
}

extern "C" int vft_linear_nf4(const void* x, void* y, const void* bias, const void* residual, const void* ext_x,
                              const void* ext_w, const void* packed, const float* absmax, const float* code,
                              int out_features, int in_features, int M, int r, int64_t ldx, int64_t ldy, int64_t ldres,
                              int transpose_w, void* workspace, int64_t ws_bytes, void* stream) {
    VFT_CHECK_ARG(x && y && packed && absmax && code, "vft_linear_nf4: null pointer");
    const int N = transpose_w ? in_features : out_features;
    const int K = transpose_w ? out_features : in_features;
    VFT_CHECK_ARG(M > 0 && N > 0 && K > 0, "vft_linear_nf4: non-positive dimension");
    VFT_CHECK_ARG(in_features % 64 == 0, "vft_linear_nf4: in_features=%d must be a multiple of 64", in_features);
    VFT_CHECK_ARG(!transpose_w || out_features % 64 == 0, "vft_linear_nf4: transposed use needs out_features %% 64 == 0 (got %d)", out_features);
    VFT_CHECK_ARG(N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "vft_linear_nf4: N, ldx, ldy must be multiples of 8");
    VFT_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)packed) & 15) == 0, "vft_linear_nf4: pointers must be 16-byte aligned");
    VFT_CHECK_ARG(bias == nullptr || ((uintptr_t)bias & 7) == 0, "vft_linear_nf4: misaligned bias");
    VFT_CHECK_ARG(residual == nullptr || (ldres % 8 == 0 && ((uintptr_t)residual & 15) == 0), "vft_linear_nf4: bad residual pointer / stride");
    VFT_CHECK_ARG((ext_x == nullptr) == (ext_w == nullptr), "vft_linear_nf4: ext_x and ext_w go together");
    VFT_CHECK_ARG(ext_x == nullptr || ((r == 4 || r == 8 || r == 16 || r == 32) && (((uintptr_t)ext_x | (uintptr_t)ext_w) & 15) == 0),
                  "vft_linear_nf4: rank extension needs r in {4, 8, 16, 32} and 16-byte aligned operands (r=%d)", r);
    LinParams p{};
    p.x = (const bf16_t*)x, p.y = (bf16_t*)y, p.bias = (const bf16_t*)bias, p.res = (const bf16_t*)residual;
    p.ex = (const bf16_t*)ext_x, p.ew = (const bf16_t*)ext_w;
    p.M = M, p.N = N, p.K = K, p.r = ext_x ? r : 0, p.ldx = ldx, p.ldy = ldy, p.ldres = ldres;
    p.q_packed = (const uint8_t*)packed, p.q_absmax = absmax, p.q_code = code, p.QN = out_features, p.QK = in_features;
    hipStream_t st = (hipStream_t)stream;
    if (workspace != nullptr) {
        // dequantise-once form on the second-generation kernel (linear8.hip: nf4_fill_and_meet): the workspace holds 4 barrier
        // words (in its first 16 KiB) and, behind them, the bf16 image of the weight for the duration of this launch
        VFT_CHECK_ARG(ws_bytes >= vft_linear_nf4_workspace(out_features, in_features) && ((uintptr_t)workspace & 255) == 0,
                      "vft_linear_nf4: workspace of %lld bytes (256-byte aligned) needed, got %lld", (long long)vft_linear_nf4_workspace(out_features, in_features), (long long)ws_bytes);
        Lin8Params q{};
        q.x = p.x, q.w = (const bf16_t*)((char*)workspace + 16384), q.y = p.y, q.bias = p.bias, q.res = p.res, q.ex = p.ex, q.ew = p.ew;
        q.M = M, q.N = N, q.K = K, q.r = p.r, q.ldx = ldx, q.ldw = K, q.ldy = ldy, q.ldres = ldres;
        q.q_packed = p.q_packed, q.q_absmax = absmax, q.q_code = code, q.QN = out_features, q.QK = in_features, q.q_trans = transpose_w ? 1 : 0;
        q.q_sync = reinterpret_cast<unsigned*>(workspace);  // at a FIXED place: layers of different sizes share one workspace
        {  // timing experiments only (read per call so that a script can switch it between calls)
            const char* e = getenv("VFT_NF4_DEBUG");
            q.q_debug = e ? atoi(e) : 0;
        }
        if (g_variant >= 0 && K % 64 == 0 && out_features % 8 == 0 && q.r <= 16 && vft_lin8_fits(q, false)) {
            launch_lin8_cfg(q, g_variant > 0 ? g_variant : pick_tile(M, N, K, 0), false, st);
            VFT_CHECK_LAUNCH("vft_linear_nf4");
            return 0;
        }
        // shapes the second-generation kernel does not take fall through to the in-loop form below
    }
    const int rem = N % 256;
    const bool wide = N >= 1024 && (rem == 0 || rem > 192) && (!transpose_w || in_features >= 32);
    if (!transpose_w) {
        if (wide)
            launch_linear<Cfg<256, 256, 2, 2, 2, false, false, 64, 1>>(p, st);
        else
            launch_linear<Cfg<256, 128, 4, 3, 2, false, false, 64, 1>>(p, st);
    } else {
        if (wide)
            launch_linear<Cfg<256, 256, 2, 2, 2, false, false, 64, 2>>(p, st);
        else
            launch_linear<Cfg<256, 128, 4, 3, 2, false, false, 64, 2>>(p, st);
    }
    VFT_CHECK_LAUNCH("vft_linear_nf4");
    return 0;
}
