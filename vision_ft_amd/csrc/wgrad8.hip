// Weight-gradient GEMM ("wg8"): bf16 operands, fp32 accumulate, gfx950.
//
//   dW[R, Cc] (+)= dy[M, R]ᵀ · x[M, Cc]                      (nn.Linear: R = out_features, Cc = in_features)
//   dW[R, 9·Cin] (+)= dy[pixels, R]ᵀ · im2col3x3(x)[pixels, 9·Cin]   (3x3 convolution, NHWC, implicit im2col)
//
// This is what autograd computes for a trainable weight behind F.linear / F.conv2d (the full fine-tune, BASELINE
// config 2: src/models/sdxl/denoiser.py:62-64,135-137,183-204 and 511-551 of the reference run it through
// torch.mm / cudnn): the reduction runs over the TOKEN (pixel) index, which is the slow index of both operands, and
// the output is small (640² … 10240×1280) against a reduction length of 16 k – 393 k rows.  Hence:
//
//   * same 8-phase main loop as lin8 (linear8.hip): 256x256 output tile, 8 waves, v_mfma_f32_16x16x32_bf16, reduction in
//     tiles of 64 rows, each staged as four half-tiles (A0, B0, B1, A1 = 64 rows x 128 columns of dy / x) by LDS-DMA
//     through buffer descriptors, three half-tiles in flight, the two 4-wave groups one barrier apart;
//   * the operands are [row][column] with the reduction index on the rows, so the MFMA fragments (8 consecutive reduction
//     elements per lane) are COLUMN reads of the staged tile: ds_read_b64_tr_b16 (hardware transpose), two per fragment.
//     The reduction index inside a 32-row step may be permuted freely as long as both operands use the same
//     permutation: lane group g takes rows {4g..4g+3, 16+4g..16+4g+3}, so the eight 32-byte row segments one
//     32-lane half reads in one instruction come from rows r with distinct r & 7;
//   * LDS image: 256-byte rows, the 32-byte segment s of row r at segment s ^ (r & 7) (the swizzle sits on the DMA's
//     per-lane SOURCE chunk: a DMA writes lane-linear) — conflict-free for those transposed reads;
//   * split over the reduction: work item = (output tile, row slice); persistent workgroups walk the items slice-major
//     (concurrent workgroups read the same rows).  Every item writes its fp32 partial tile; a second kernel adds the
//     slices in slice order (+ the previous gradient when accumulating) — no atomics, bit-reproducible;
//   * rows beyond M and columns beyond R / Cc read zeros through the descriptors' bounds check (the row advance is part
//     of the per-lane offset, not of the scalar offset, so that the check sees it).
#include <type_traits>

#include "vft_common.h"
#include "wgrad8.h"

namespace {

constexpr int NT = 512;
constexpr int BM = 256, BN = 256;        // output tile: rows from dy's columns, columns from x's columns
constexpr int FX = 4, FW = 2;            // 16-wide fragments per wave and half-tile (wave tile 128 x 64, 2 x 4 waves)
constexpr int HALF = 64 * 256;           // bytes of one half-tile: 64 reduction rows x 128 columns
constexpr int TILE = 4 * HALF;           // one reduction tile: A0 A1 B0 B1
constexpr int DX = 2, DW = 2;            // DMA instructions per wave per half-tile (1 KiB = 4 rows each)
constexpr size_t LDS_BYTES = 2 * (size_t)TILE;
constexpr int SLOTS = 256;               // resident workgroups (one per CU)
constexpr int OOB = (int)0x80000000u;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Two LDS-DMA pieces (2 x 1 KiB, consecutive LDS destinations) in ONE inline-asm statement.  Through the builtin the compiler
// sees a pending LDS write on the VM counter and puts `s_waitcnt vmcnt(0)` in front of the next ds_read_b64_tr_b16 — in
// this kernel that was EVERY phase of the main loop (eight vmcnt(0) per reduction tile in the round-2 ISA): the three
// half-tiles "in flight" were drained as soon as they were issued.  Hidden in asm the DMAs are outside that bookkeeping
// and the counted waits written below are the only ones (see attn.hip for the M0 / s_nop rules of the statement).
__device__ __forceinline__ void dma16x2(__amdgpu_buffer_rsrc_t r, unsigned char* l, int voff0, int voff1) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)l);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %2, 0 offen lds\n\t"
                 "s_add_u32 m0, %1, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(dst), "s"(r), "v"(voff0), "v"(voff1)
                 : "scc");
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ bf16x8 tr8(const unsigned char* a) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * 256));
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

struct Wg8Params {
    const bf16_t* a;  // dy [M][lda]
    const bf16_t* b;  // x  [M][ldb]      (CONV: NHWC activation [images*H*W][ldb = Cin])
    float* part;      // [slices][R][Cc] fp32 partial sums
    int M, R, Cc;
    int64_t lda, ldb;
    int a_bytes, b_bytes;  // extents for the buffer descriptors (< 2^31)
    int slices;
    int cH, cW, cC;  // CONV: image height / width / input channels (Cc = 9 * cC)
    // batched mode (nbatch > 1; attention's dQ = dS^T-transposed x K per (batch, head), see vft_wg8_batched): operand bases of
    // batch bi = (bo, bi_in) with bo = bi / inner, bi_in = bi % inner:  a + bi * a_bs,  b + bo * b_os + (bi_in / G) * b_is
    int nbatch, inner, G;
    int64_t a_bs, b_os, b_is;
    // batched mode with ONE slice: the scaled bf16 result straight from the accumulators (no partial tiles, no reduce pass)
    bf16_t* direct;
    int64_t ldo, o_os, o_is;
    float oscale;
};

template <bool CONV>
__global__ __launch_bounds__(NT, 2) void wg8_kernel(Wg8Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3, grp = wv >> 2;

    // ---- work items: (row slice, output tile), slice-major.  Output columns are handled in halves of 128: a half never
    //      straddles a convolution tap (each tap's Cin is rounded up to whole halves; the overhang reads zeros)
    const int hpt = ((CONV ? p.cC : p.Cc) + 127) / 128, nhalves = (CONV ? 9 : 1) * hpt;
    const int tiles_r = (p.R + BM - 1) / BM, tiles_c = (nhalves + 1) / 2, ntiles = tiles_r * tiles_c;
    const int nitems = ntiles * p.nbatch * p.slices, nkt_all = (p.M + 63) / 64;
    int r0, h0, kt0, nkt, slice, batch;
    // id -> item: XCD-contiguous (workgroup ids go round-robin over the 8 XCDs and the grid is a multiple of 8 or a single
    // round, so vt & 7 labels the XCD): one XCD takes whole slices, whose tiles share their dy / x column blocks through
    // that XCD's L2 (measured: FETCH_SIZE of the 1280 x 1280 x 16384 problem 420 MB with round-robin items)
    auto decode = [&](int vt) {
        const int xcd = vt & 7, q = nitems >> 3, rr = nitems & 7;
        const int lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (vt >> 3);
        slice = lin / (ntiles * p.nbatch);
        const int rem = lin - slice * ntiles * p.nbatch;
        batch = rem / ntiles;
        const int tile = rem - batch * ntiles;
        r0 = (tile / tiles_c) * BM;
        h0 = (tile % tiles_c) * 2;
        kt0 = (int)((int64_t)slice * nkt_all / p.slices);
        nkt = (int)((int64_t)(slice + 1) * nkt_all / p.slices) - kt0;
    };

    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.b_bytes, 0x00020000);
    const int stepa = (int)(64 * p.lda * 2), stepb = (int)(64 * p.ldb * 2);  // bytes per reduction tile

    // ---- LDS-DMA sources.  One instruction = 4 rows x 256 B: lane -> (row = lane >> 4, slot = lane & 15); the slot holds
    //      source chunk slot ^ ((row & 7) << 1)
    int ga[2][DX], gb[2][DW];  // byte offsets of this lane's chunks at the slice's first reduction tile
    // CONV: the 8 pixel rows a wave stages per reduction tile (rows 8 wv .. 8 wv + 7 of the tile) lie in ONE image row
    // (W % 8 == 0), so the image coordinates of the wave's first pixel are wave-uniform and advance on the scalar unit
    // (64 pixels per reduction tile); a lane adds its 0..7.  Taps are per half: (tdy, tdx) of the half's tap.
    int s_oh = 0, s_ow = 0, tdy[2] = {0, 0}, tdx[2] = {0, 0};
    const int q64 = CONV ? 64 / p.cW : 0, r64 = CONV ? 64 % p.cW : 0;
    const int lrow = lane >> 4;
    auto setup = [&]() {
        if (p.nbatch > 1) {  // this item's operands (wave-uniform: scalar descriptor rebuild)
            const int bo = batch / p.inner, bin = batch - bo * p.inner;
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (int64_t)batch * p.a_bs), 0, p.a_bytes, 0x00020000);
            rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + (int64_t)bo * p.b_os + (int64_t)(bin / p.G) * p.b_is), 0, p.b_bytes, 0x00020000);
        }
        if (CONV) {
            const int m0 = kt0 * 64 + 8 * wv, pix = m0 % (p.cH * p.cW);
            s_oh = __builtin_amdgcn_readfirstlane(pix / p.cW);
            s_ow = __builtin_amdgcn_readfirstlane(pix % p.cW);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int tap = (h0 + hh) / hpt;
                tdy[hh] = tap / 3 - 1, tdx[hh] = tap % 3 - 1;
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int q = 0; q < DX; ++q) {
                const int r = (wv * DX + q) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ ((r & 7) << 1);
                const int col = r0 + hh * 128 + ch * 8;
                ga[hh][q] = col < p.R ? (int)((((int64_t)kt0 * 64 + r) * p.lda + col) * 2) : OOB;
                const int h = h0 + hh;
                const int tap = CONV ? h / hpt : 0, c0 = (CONV ? h % hpt : h) * 128;
                const int cb = c0 + ch * 8;
                const bool ok = h < nhalves && cb < (CONV ? p.cC : p.Cc);
                const int shift = CONV ? (tap / 3 - 1) * p.cW + (tap % 3 - 1) : 0;
                gb[hh][q] = ok ? (int)((((int64_t)kt0 * 64 + r + shift) * p.ldb + cb) * 2) : OOB;
            }
        }
    };
    auto stage_a = [&](int t, int buf, int hh) {
        unsigned char* dst = smw + buf * TILE + hh * HALF + wv * DX * 1024;
        const int adv = t * stepa;
        static_assert(DX == 2 && DW == 2, "dma16x2 issues two pieces");
        dma16x2(ra, dst, (int)((unsigned)ga[hh][0] + (unsigned)adv), (int)((unsigned)ga[hh][1] + (unsigned)adv));
    };
    auto stage_b = [&](int t, int buf, int hh) {
        unsigned char* dst = smw + buf * TILE + 2 * HALF + hh * HALF + wv * DW * 1024;
        const int adv = t * stepb;
        if (CONV) {
            // the two halves of a reduction tile are always staged half 0 first, tiles in order: advance with half 0
            if (hh == 0 && t != 0) {
                s_ow += r64, s_oh += q64;
                if (s_ow >= p.cW) s_ow -= p.cW, s_oh += 1;
                while (s_oh >= p.cH) s_oh -= p.cH;
            }
            const bool rowok = (unsigned)(s_oh + tdy[hh]) < (unsigned)p.cH;
            int vo[DW];
#pragma unroll
            for (int q = 0; q < DW; ++q) {
                const bool ok = rowok && (unsigned)(s_ow + 4 * q + lrow + tdx[hh]) < (unsigned)p.cW;
                vo[q] = ok ? (int)((unsigned)gb[hh][q] + (unsigned)adv) : OOB;
            }
            dma16x2(rb, dst, vo[0], vo[1]);
        } else {
            dma16x2(rb, dst, (int)((unsigned)gb[hh][0] + (unsigned)adv), (int)((unsigned)gb[hh][1] + (unsigned)adv));
        }
    };

    // ---- transposed fragment reads: lane group g = lane >> 4 takes rows 4g + q (+16 for the second read), q = (lane & 15) >> 2,
    //      8-byte piece pp = lane & 3 of the 32-byte segment; segment s of row r sits at s ^ (r & 7), r & 7 = 4 (g & 1) + q
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int fsw = 4 * (g4 & 1) + q4;
    const int rowoff = (4 * g4 + q4) * 256 + 8 * pp;
    int xa[FX], wa[FW];
#pragma unroll
    for (int i = 0; i < FX; ++i) xa[i] = rowoff + (((wr * 4 + i) ^ fsw) << 5);
#pragma unroll
    for (int j = 0; j < FW; ++j) wa[j] = 2 * HALF + rowoff + (((wc * 2 + j) ^ fsw) << 5);

    f32x4 acc[2][FX][2][FW];
    bf16x8 xf[FX][2], w0[FW][2], w1[FW][2];
    auto read_x = [&](const unsigned char* tb, int hh) {
#pragma unroll
        for (int i = 0; i < FX; ++i) {
            xf[i][0] = tr8(tb + hh * HALF + xa[i]);
            xf[i][1] = tr8(tb + hh * HALF + 32 * 256 + xa[i]);
        }
    };
    auto read_w = [&](const unsigned char* tb, int hh, bf16x8 (&wf)[FW][2]) {
#pragma unroll
        for (int j = 0; j < FW; ++j) {
            wf[j][0] = tr8(tb + hh * HALF + wa[j]);
            wf[j][1] = tr8(tb + hh * HALF + 32 * 256 + wa[j]);
        }
    };
    auto mma = [&](f32x4 (&a)[FX][2][FW], int hw, const bf16x8 (&wf)[FW][2]) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < FX; ++i)
#pragma unroll
                for (int j = 0; j < FW; ++j) a[i][hw][j] = mfma16(wf[j][kk], xf[i][kk], a[i][hw][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto lgk0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    // One reduction tile = four phases (the schedule of lin8: see linear8.hip)
    // two phases of 32 MFMAs per wave and reduction tile (lin8's round-3 schedule: half as many hand-overs between the groups)
    auto tile = [&](int t, auto BUFc, bool has1, bool has2) {
        constexpr int BUF = decltype(BUFc)::value;
        const unsigned char* tb = smw + BUF * TILE;
        read_x(tb, 0);
        read_w(tb, 0, w0);
        read_w(tb, 1, w1);
        if (has1) stage_a(t + 1, BUF ^ 1, 1);
        lgk0();
        bar();
        mma(acc[0], 0, w0);
        mma(acc[0], 1, w1);
        bar();
        read_x(tb, 1);
        if (has2) {
            stage_a(t + 2, BUF, 0);
            stage_b(t + 2, BUF, 0);
            stage_b(t + 2, BUF, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DX + 2 * DW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lgk0();  // X1's slot is re-filled by the first phase of the next tile: its reads are done before this rendezvous
        bar();
        mma(acc[1], 1, w1);
        mma(acc[1], 0, w0);
        bar();
    };
    auto prologue = [&]() {
        stage_a(0, 0, 0);
        stage_b(0, 0, 0);
        stage_b(0, 0, 1);
        stage_a(0, 0, 1);
        if (nkt > 1) {
            stage_a(1, 1, 0);
            stage_b(1, 1, 0);
            stage_b(1, 1, 1);
        }
    };

    int vt = blockIdx.x;
    decode(vt);
    setup();
    prologue();
    for (;;) {
        const int cnkt = nkt;
        if (cnkt > 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DX + 2 * DW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
        if (grp == 1) bar();  // stagger the groups by one barrier
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < FX; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < FW; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[a][i][b][j][e] = 0.f;
        {
            int t = 0;
            for (; t + 1 < cnkt; t += 2) {
                tile(t, std::integral_constant<int, 0>{}, true, t + 2 < cnkt);
                tile(t + 1, std::integral_constant<int, 1>{}, t + 2 < cnkt, t + 3 < cnkt);
            }
            if (t < cnkt) tile(t, std::integral_constant<int, 0>{}, false, false);
        }
        if (grp == 0) bar();  // both groups are past their last fragment reads: the buffers are free

        // ---- the next item's first DMAs fly while this item's partial tile is written out ----------------------
        const int cr0 = r0, ch0 = h0, cslice = slice, cbatch = batch;
        const int vn = vt + gridDim.x;
        const bool more = vn < nitems;
        if (more) {
            decode(vn);
            setup();
            prologue();
        }
        int etid = tid;  // tail addresses must not stay live across the main loop (see linear8.hip)
        asm volatile("" : "+v"(etid));
        const int el15 = etid & 15, elq = (etid & 63) >> 4;
        float* const out = p.part + ((int64_t)cslice * p.nbatch + cbatch) * p.R * p.Cc;
        bf16_t* dout = nullptr;
        if (!CONV && p.direct) {
            const int bo = cbatch / p.inner;
            dout = p.direct + (int64_t)bo * p.o_os + (int64_t)(cbatch - bo * p.inner) * p.o_is;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int h = ch0 + b;
            const int tap = CONV ? h / hpt : 0, c0 = (CONV ? h % hpt : h) * 128;
            const int lim = CONV ? p.cC : p.Cc, cbase = (CONV ? tap * p.cC : 0) + c0;
#pragma unroll
            for (int j = 0; j < FW; ++j) {
                const int within = wc * 32 + j * 16 + 4 * elq;
                const bool cok = h < nhalves && c0 + within < lim;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < FX; ++i) {
                        const int row = cr0 + a * 128 + wr * 64 + i * 16 + el15;
                        if (cok && row < p.R) {
                            if (!CONV && dout) {
                                bf16x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc[a][i][b][j][e] * p.oscale);
                                *reinterpret_cast<bf16x4*>(dout + (int64_t)row * p.ldo + cbase + within) = v;
                            } else {
                                *reinterpret_cast<f32x4*>(out + (int64_t)row * p.Cc + cbase + within) = acc[a][i][b][j];
                            }
                        }
                    }
            }
        }
        if (!more) break;
        vt = vn;
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum over slices (slice order) of part[s][i]; 4 elements per thread
struct Wg8Reduce {
    const float* part;
    void* out;
    int64_t n;      // R * Cc
    int Cc;
    int64_t ldo;    // row stride of out (elements)
    int slices, accumulate, out_f32;
};
__global__ __launch_bounds__(256) void wg8_reduce_kernel(Wg8Reduce p) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= p.n) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(p.part + i);
    for (int k = 1; k < p.slices; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.part + (int64_t)k * p.n + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += t[e];
    }
    const int64_t row = i / p.Cc, col = i - row * p.Cc, o = row * p.ldo + col;
    if (p.out_f32) {
        float* q = (float*)p.out + o;
        if (p.accumulate) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += t[e];
        }
        *reinterpret_cast<f32x4*>(q) = s;
    } else {
        bf16_t* q = (bf16_t*)p.out + o;
        if (p.accumulate) {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(q);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (float)t[e];
        }
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)s[e];
        *reinterpret_cast<bf16x4*>(q) = v;
    }
}

// Batched form of the reduce pass: out[bi][row][col] (bf16, strided per batch) = scale * sum over slices of part[s][bi][row][col]
struct Wg8ReduceB {
    const float* part;
    bf16_t* out;
    int64_t n;       // R * Cc
    int Cc, slices, nbatch, inner;
    int64_t ldo, o_os, o_is;
    float scale;
};
__global__ __launch_bounds__(256) void wg8_reduce_batched_kernel(Wg8ReduceB p) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int bi = blockIdx.y;
    if (i >= p.n) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(p.part + (int64_t)bi * p.n + i);
    for (int k = 1; k < p.slices; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.part + ((int64_t)k * p.nbatch + bi) * p.n + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += t[e];
    }
    const int64_t row = i / p.Cc, col = i - row * p.Cc;
    const int bo = bi / p.inner, bin = bi - bo * p.inner;
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(s[e] * p.scale);
    *reinterpret_cast<bf16x4*>(p.out + (int64_t)bo * p.o_os + (int64_t)bin * p.o_is + row * p.ldo + col) = v;
}

// Row slices: the count that minimises  rounds x (reduction tiles per slice + fixed per-item cost)  on 256 workgroups
int pick_slices(int64_t M, int R, int halves) {
    const int ntiles = ((R + BM - 1) / BM) * ((halves + 1) / 2);
    const int nkt = (int)((M + 63) / 64);
    int best = 1;
    double bc = 1e30;
    const double item_cost = 24.0;  // prologue + fp32 partial tile + its share of the reduce pass, in reduction steps (swept 4-24 on the SDXL shapes)
    for (int s = 1; s <= 64 && s * 4 <= nkt; ++s) {
        const int rounds = (ntiles * s + SLOTS - 1) / SLOTS;
        const double c = rounds * ((double)((nkt + s - 1) / s) + item_cost) + 0.35 * s;  // last term: the reduce pass reads s partials
        if (c < bc) bc = c, best = s;
    }
    return best;
}

template <bool CONV>
void launch_wg8(const Wg8Params& p, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(wg8_kernel<CONV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_BYTES);
        attr = true;
    }
    const int hpt = ((CONV ? p.cC : p.Cc) + 127) / 128, nhalves = (CONV ? 9 : 1) * hpt;
    const unsigned items = (unsigned)(((p.R + BM - 1) / BM) * ((nhalves + 1) / 2) * p.slices * p.nbatch);
    const unsigned grid = items < (unsigned)SLOTS ? items : (unsigned)SLOTS;
    hipLaunchKernelGGL(wg8_kernel<CONV>, dim3(grid), dim3(NT), LDS_BYTES, st, p);
}

int run_wgrad(const void* dy, const void* x, void* dw, float* scratch, int64_t M, int R, int Cc, int64_t lddy, int64_t ldx,
              int64_t lddw, int accumulate, int out_f32, int cH, int cW, int cC, void* stream, const char* name) {
    const bool conv = cC > 0;
    VFT_CHECK_ARG(dy && x && dw && scratch, "%s: null pointer", name);
    VFT_CHECK_ARG(M > 0 && R > 0 && Cc > 0 && R % 8 == 0 && Cc % 8 == 0, "%s: unsupported shape M=%lld R=%d Cc=%d", name,
                  (long long)M, R, Cc);
    VFT_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0 && lddw % 4 == 0 && lddy >= R && lddw >= Cc, "%s: strides", name);
    VFT_CHECK_ARG((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) & 15) == 0, "%s: misaligned pointer", name);
    const int64_t xcols = conv ? cC : Cc;
    VFT_CHECK_ARG(ldx >= xcols, "%s: strides", name);
    VFT_CHECK_ARG(!conv || (Cc == 9 * cC && cH > 0 && cW > 0 && cW % 8 == 0 && M % ((int64_t)cH * cW) == 0),
                  "%s: convolution geometry (W %% 8 == 0)", name);
    const int64_t ab = ((M - 1) * lddy + R) * 2, bb = ((M - 1) * ldx + xcols) * 2;
    // the per-lane offsets (row advance included) are 32-bit; 64 rows of slack for the last, partial reduction tile
    VFT_CHECK_ARG(ab + 130 * lddy < (int64_t)1 << 31 && bb + 130 * ldx < (int64_t)1 << 31, "%s: operand beyond 2 GiB", name);
    VFT_CHECK_ARG(M < (int64_t)1 << 30, "%s: M", name);
    Wg8Params p{};
    p.a = (const bf16_t*)dy, p.b = (const bf16_t*)x, p.part = scratch, p.M = (int)M, p.R = R, p.Cc = Cc, p.lda = lddy, p.ldb = ldx;
    p.a_bytes = (int)ab, p.b_bytes = (int)bb, p.cH = cH, p.cW = cW, p.cC = cC;
    p.nbatch = 1, p.inner = 1, p.G = 1;
    const int halves = (conv ? 9 : 1) * (int)((xcols + 127) / 128);
    p.slices = pick_slices(M, R, halves);
    if (conv) launch_wg8<true>(p, (hipStream_t)stream);
    else launch_wg8<false>(p, (hipStream_t)stream);
    Wg8Reduce q{};
    q.part = scratch, q.out = dw, q.n = (int64_t)R * Cc, q.Cc = Cc, q.ldo = lddw, q.slices = p.slices, q.accumulate = accumulate;
    q.out_f32 = out_f32;
    hipLaunchKernelGGL(wg8_reduce_kernel, dim3((unsigned)((q.n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q);
    return 0;
}

}  // namespace

// ---- internal: batched transposed-A product with a scaled bf16 result (attn.hip: dQ = scale * dS K per (batch, head)) ----
static int batched_slices(int64_t M, int R, int Cc, int nbatch) {
    // items = batches x tiles: with many batches one round-robin over the 256 workgroups is balanced without a reduction split
    const int ntiles = ((R + BM - 1) / BM) * ((((Cc + 127) / 128) + 1) / 2);
    return (int64_t)ntiles * nbatch >= 2 * SLOTS ? 1 : pick_slices(M, R, (Cc + 127) / 128);
}
int64_t vft_wg8_batched_scratch_floats(int64_t M, int R, int Cc, int nbatch) {
    const int ns = batched_slices(M, R, Cc, nbatch);
    return ns == 1 ? 4 : (int64_t)ns * nbatch * R * Cc;  // one slice: the result is written directly
}
int vft_wg8_batched(const void* a, const void* b, void* out, float* scratch, int64_t M, int R, int Cc, int64_t lda, int64_t ldb,
                    int64_t ldo, const Wg8Batch& bt, hipStream_t stream) {
    const char* name = "vft_wg8_batched";
    VFT_CHECK_ARG(a && b && out && scratch && M > 0 && R > 0 && Cc > 0 && R % 8 == 0 && Cc % 8 == 0 && bt.nbatch > 0 && bt.inner > 0 && bt.G > 0,
                  "%s: bad arguments", name);
    VFT_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldo % 4 == 0 && lda >= R && ldb >= Cc && ldo >= Cc, "%s: strides", name);
    const int64_t ab = ((M - 1) * lda + R) * 2, bb = ((M - 1) * ldb + Cc) * 2;
    VFT_CHECK_ARG(ab + 130 * lda < (int64_t)1 << 31 && bb + 130 * ldb < (int64_t)1 << 31 && M < (int64_t)1 << 30, "%s: operand beyond 2 GiB", name);
    Wg8Params p{};
    p.a = (const bf16_t*)a, p.b = (const bf16_t*)b, p.part = scratch, p.M = (int)M, p.R = R, p.Cc = Cc, p.lda = lda, p.ldb = ldb;
    p.a_bytes = (int)ab, p.b_bytes = (int)bb;
    p.nbatch = bt.nbatch, p.inner = bt.inner, p.G = bt.G, p.a_bs = bt.a_bstride, p.b_os = bt.b_outer, p.b_is = bt.b_inner;
    p.slices = batched_slices(M, R, Cc, bt.nbatch);
    if (p.slices == 1) {
        p.direct = (bf16_t*)out, p.ldo = ldo, p.o_os = bt.o_outer, p.o_is = bt.o_inner, p.oscale = bt.scale;
        launch_wg8<false>(p, stream);
        return 0;
    }
    launch_wg8<false>(p, stream);
    Wg8ReduceB q{};
    q.part = scratch, q.out = (bf16_t*)out, q.n = (int64_t)R * Cc, q.Cc = Cc, q.slices = p.slices, q.nbatch = bt.nbatch, q.inner = bt.inner;
    q.ldo = ldo, q.o_os = bt.o_outer, q.o_is = bt.o_inner, q.scale = bt.scale;
    hipLaunchKernelGGL(wg8_reduce_batched_kernel, dim3((unsigned)((q.n / 4 + 255) / 256), (unsigned)bt.nbatch), dim3(256), 0, stream, q);
    return 0;
}

extern "C" int64_t vft_linear_wgrad_scratch_floats(int64_t M, int R, int Cc) {
    if (M <= 0 || R <= 0 || Cc <= 0) return 0;
    return (int64_t)pick_slices(M, R, (Cc + 127) / 128) * R * Cc;
}

extern "C" int vft_linear_wgrad(const void* dy, const void* x, void* dw, float* scratch, int64_t M, int R, int Cc,
                                int64_t lddy, int64_t ldx, int64_t lddw, int accumulate, int out_f32, void* stream) {
    const int st = run_wgrad(dy, x, dw, scratch, M, R, Cc, lddy, ldx, lddw, accumulate, out_f32, 0, 0, 0, stream, "vft_linear_wgrad");
    if (st != 0) return st;
    VFT_CHECK_LAUNCH("vft_linear_wgrad");
    return 0;
}

extern "C" int64_t vft_conv3x3_wgrad_scratch_floats(int64_t pixels, int Cout, int Cin) {
    if (pixels <= 0 || Cout <= 0 || Cin <= 0) return 0;
    return (int64_t)pick_slices(pixels, Cout, 9 * ((Cin + 127) / 128)) * Cout * 9 * Cin;
}

extern "C" int vft_conv3x3_wgrad(const void* dy, const void* x, void* dw, float* scratch, int B, int H, int W, int Cin, int Cout,
                                 int accumulate, int out_f32, void* stream) {
    const int64_t M = (int64_t)B * H * W;
    const int st = run_wgrad(dy, x, dw, scratch, M, Cout, 9 * Cin, Cout, Cin, 9 * (int64_t)Cin, accumulate, out_f32, H, W, Cin, stream,
                             "vft_conv3x3_wgrad");
    if (st != 0) return st;
    VFT_CHECK_LAUNCH("vft_conv3x3_wgrad");
    return 0;
}
