// Projection GEMM, second generation ("lin8"): bf16, fp32 accumulate, gfx950.
//
//   y[M,N] = x[M,K] · W[N,K]ᵀ  (+ bias[N])  (+ ex[M,r] · ew[N,r]ᵀ)  (+ res[M,N])          K % 64 == 0
//
// Same contract as linear.hip's first-generation kernel (the reference's LoRALinear.forward,
// src/modules/peft/lora.py:92-104, as ONE launch: base GEMM + bias + rank-r update + residual add; the data
// gradient is the same call on the transposed frozen weight), and the same implicit-GEMM convolution mode
// (src/models/sdxl/denoiser.py:511-551).  What changed is the main loop, which now follows the CDNA4 playbook's
// 8-phase structure:
//
//   * 256x256 (or 256x128 / 128x128) output tile, 8 waves, K in tiles of 64 (full 128-byte lines).
//   * v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it than on 32x32x16 at equal cycles/flop),
//     weight = A operand, so an accumulator register group is 4 consecutive output features of one token.
//   * Each K-tile is staged as FOUR half-tiles (X0, W0, W1, X1: rows 0..BM/2, BM/2..BM of x, likewise W) by LDS-DMA
//     (buffer_load ... lds).  A wave's output tile takes WM/2 rows from EACH X half and WN/2 columns from each W half; a
//     K-tile is two phases of two quadrants each (round 3; four one-quadrant phases in round 2):
//         phase A  reads X0, W0, W1   quadrants (X0,W0) (X0,W1)   issues X1 of tile t+1
//         phase B  reads X1           quadrants (X1,W1) (X1,W0)   issues X0, W0, W1 of tile t+2, s_waitcnt vmcnt(those three
//                                                                 stay in flight)
//     i.e. two K-tile buffers (2 x 64 KiB at 256x256) and three half-tiles always in flight across the barriers;
//     vmcnt is never drained inside the loop.
//   * The two 4-wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same program one barrier apart,
//     so on every SIMD one wave issues its 32 MFMAs while its partner does its fragment reads and DMA issue.
//   * LDS image is lane-linear per DMA (8 rows x 128 B per wave-instruction); the bank swizzle
//     chunk' = chunk ^ (row & 7) is applied to the per-lane SOURCE chunk and again on the ds_read_b128 fragment
//     reads (conflict-free for the 16x16x32 operand: 16 rows x 4 k-chunks per instruction).
//   * XCD-aware tile order, epilogue through LDS (full 512-byte row segments), rank extension as one extra MFMA
//     k-step from registers.
#include <cstdlib>
#include <type_traits>

#include "vft_common.h"
#include "linear8.h"

// Diagnostic build only (tools/exp/lin8_probe.hip defines L8_STAMPS): cycle stamps of the first tiles of every
// workgroup, written by one lane.  Never compiled into libvft_hip.so.
#ifdef L8_STAMPS
__device__ unsigned long long g_l8_stamps[512][8][12];
#define L8_STAMP(i)                                                                              \
    do {                                                                                         \
        if (tid == 0 && stamp_it < 8) g_l8_stamps[blockIdx.x][stamp_it][i] = (unsigned long long)clock64(); \
    } while (0)
#else
#define L8_STAMP(i) \
    do {            \
    } while (0)
#endif

// outputs above this size are stored with the non-temporal hint (see the epilogue)
#ifndef L8_NT_BYTES
#define L8_NT_BYTES (16ll << 20)
#endif

namespace {

constexpr int NT = 512;

typedef __attribute__((address_space(3))) void lds_void_t;

// LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds): the per-lane source is a 32-bit byte offset
// (half the registers of a 64-bit pointer, no 64-bit address arithmetic per stage), the K advance rides in the scalar
// offset, and a lane whose offset lies beyond the descriptor's extent delivers ZEROS — the convolution's padding.
constexpr int OOB = (int)0x80000000u;  // beyond every extent this kernel accepts (< 2^31 bytes)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* l, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

template <int BM_, int BN_, int WAVES_M_, bool CONV_, int STRIDE_ = 1>
struct C8 {
    static constexpr bool CONV = CONV_;
    static constexpr int STRIDE = STRIDE_;  // CONV: 1 = output grid == input grid; 2 = stride-2 / padding-1 (the SDXL Downsample)
    static_assert(STRIDE_ == 1 || (CONV_ && STRIDE_ == 2), "stride 2 is a convolution mode");
    static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = 8 / WAVES_M_;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile
    static constexpr int FX = WM / 32, FW = WN / 32;            // 16-wide fragments per sub-tile (half of the wave tile)
    static constexpr int XH = BM / 2 * 128, WH = BN / 2 * 128;  // bytes per half-tile (rows x 64 bf16)
    static constexpr int TILE = 2 * XH + 2 * WH;                // one K-tile
    static constexpr int DX = BM / 128, DW = BN / 128;          // DMA instructions per wave per half-tile
    static constexpr int EP_ROWS = 64;                          // token rows per epilogue pass
    static constexpr int EP_BYTES = EP_ROWS * BN * 2;           // epilogue image: its own region, so that the next
                                                                // tile's DMAs can fly while this tile is written out
    static constexpr size_t LDS_BYTES = 2 * (size_t)TILE + EP_BYTES;
    static constexpr int WPS = LDS_BYTES <= 80 * 1024 ? 4 : 2;  // waves per SIMD the launch bound promises
    static constexpr int SLOTS = (LDS_BYTES <= 80 * 1024 ? 2 : 1) * 256;  // resident workgroups on the chip
    static_assert(WM % 32 == 0 && WN % 32 == 0 && WM / 2 <= EP_ROWS, "wave tile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// ---- NF4 base weight: dequantise ONCE per launch, inside the launch ------------------------------------------------
// (replaces bnb.matmul_4bit of BnbLinear4bit, src/modules/quant/bnb.py:37-129.)  Round 3 dequantised inside the K-loop:
// every row tile decoded the same weights again (68 times at M = 17 408) and the table look-ups saturated the loop's LDS
// pipe.  Here every workgroup of the persistent grid first converts 1/grid of the packed codes into a bf16
// scratch, the grid meets at a barrier, and the main loop then runs exactly as for a bf16 weight.  The values are those
// of vft_nf4_dequant, bit for bit: bf16(code[c] * absmax).
//   forward (q_trans = 0):        scratch[n][k] = W[n][k]   — elementwise, 8 weights (4 code bytes -> 16 bytes) per step
//   data gradient (q_trans = 1):  scratch[i][o] = W[o][i]   — blocks of 128 o x 64 i transposed through LDS: a thread decodes
//       8 consecutive i of rows o, o + 1 and writes (o, o+1) pairs as dwords into a [64 i][128 o] image (row stride 260 B:
//       the 64 lanes of a write hit 64 banks), then rows of the image leave as 16-byte pieces.
// Barrier: one relaxed atomic per workgroup into one of eight arrival words (q_sync[64 + 16 c]) that are never reset, a
// launch counter (q_sync[0]) that workgroup 0 advances; the cache maintenance that makes the image visible across the eight
// XCDs' L2s is explicit, once per workgroup; the spin is BOUNDED (every wave reaches the end of the kernel whatever
// happens: on a time-out the error word q_sync[2] is set and the host-side tests look at it).  No host-side memset between
// launches (graph replay).  Co-residency of the grid is the launcher's duty (grid <= occupancy x CUs, launch8k).
// 16-byte store that is WRITTEN THROUGH this XCD's L2 (sc0 sc1): the image of the weight must be visible to the other seven
// XCDs behind the barrier; written through, nothing of it is a dirty L2 line when the workgroups meet and the release side
// of the barrier needs no `buffer_wbl2` sweep.  (tools/exp/nf4_time.py, M 17408 N 3840 K 2304 on a warm chip: the launch
// without conversion and barrier 210 us = the bf16 launch; conversion +8; barrier +14; both 235; plain stores + one
// `buffer_wbl2` per workgroup 235 as well; a separate dequantisation kernel in front of the bf16 launch 225.)
__device__ __forceinline__ void store16_through(void* dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
}

// Returns true (workgroup-uniform) when this workgroup gave up at the barrier: the caller must NOT run the main loop on
// the incomplete image — it poisons its output tiles with NaN instead (lin8_kernel) and the sticky error word q_sync[2]
// is what Trainer / bench / tests read (kernels.nf4_barrier_timed_out).
__device__ __forceinline__ bool nf4_fill_and_meet(const Lin8Params& p, unsigned char* lds) {
    const int tid = threadIdx.x;
    // (VFT_NF4_DEBUG & 16, tools/exp/nf4_time.py: cycle stamps of every workgroup at start / conversion done / barrier passed;
    // 32 bytes per workgroup from byte 4096 of the 16 KiB header: workgroups 384 and up have no slot)
    unsigned long long* stamps = ((p.q_debug & 16) && blockIdx.x < 384) ? reinterpret_cast<unsigned long long*>(p.q_sync) + 512 + 4 * blockIdx.x : nullptr;
    if (stamps && tid == 0) stamps[0] = wall_clock64();
    const unsigned epoch = __hip_atomic_load(&p.q_sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // launches so far (see the barrier)
    float* code_s = reinterpret_cast<float*>(lds + 64 * 260);
    unsigned* fail_s = reinterpret_cast<unsigned*>(lds + 64 * 260 + 64);
    if (tid < 16) code_s[tid] = p.q_code[tid];
    if (tid == 16) *fail_s = 0u;
    __syncthreads();
    bf16_t* const out = const_cast<bf16_t*>(p.w);
    auto decode = [&](unsigned d, float am, bf16_t (&v)[8]) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const unsigned byte = (d >> (8 * bb)) & 0xffu;
            v[2 * bb] = (bf16_t)(code_s[byte >> 4] * am);
            v[2 * bb + 1] = (bf16_t)(code_s[byte & 15] * am);
        }
    };
    if (p.q_debug & 1) {
    } else if (!p.q_trans) {
        // one step = 4 code bytes -> 8 weights -> one 16-byte store; consecutive lanes take consecutive steps (every load
        // instruction 256 contiguous bytes, every store 1 KiB: the first version gave a lane 16 code bytes and its four
        // stores landed 64 bytes apart from the neighbour lane's — the conversion then took ~80 us instead of ~10), and a
        // thread requests EIGHT steps, a grid-width apart, before it decodes the first: one memory round trip for the
        // whole weight at Lumina2's sizes (8.4 steps per thread)
        const int64_t nvec = (int64_t)p.QN * p.QK / 8, T = (int64_t)gridDim.x * NT;
        for (int64_t e0 = (int64_t)blockIdx.x * NT + tid; e0 < nvec; e0 += 8 * T) {
            unsigned d[8];
            float am[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t e = e0 + u * T;
                d[u] = 0u, am[u] = 0.f;
                if (e < nvec) {
                    d[u] = reinterpret_cast<const unsigned*>(p.q_packed)[e];
                    am[u] = p.q_absmax[e >> 3];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t e = e0 + u * T;
                if (e < nvec) {
                    bf16_t v[8];
                    decode(d[u], am[u], v);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = v[j];
                    store16_through(out + e * 8, __builtin_bit_cast(u32x4, o));
                }
            }
        }
    } else {
        const int nbi = p.QK / 64, nblk = ((p.QN + 127) / 128) * nbi;
        const int op = tid >> 3, ich = tid & 7;
        // the codes of block `blk` for this thread (rows o, o + 1; 8 consecutive i): requested one block AHEAD of their use
        auto fetch = [&](int blk, unsigned& d0, unsigned& d1, float& a0, float& a1) {
            d0 = d1 = 0u, a0 = a1 = 0.f;
            if (blk >= nblk) return;
            const int bo = blk / nbi, i0 = (blk - bo * nbi) * 64, o = bo * 128 + 2 * op;
            if (o < p.QN) {
                const int64_t e0 = (int64_t)o * p.QK + i0;
                d0 = *reinterpret_cast<const unsigned*>(p.q_packed + (e0 >> 1) + ich * 4), a0 = p.q_absmax[e0 >> 6];
            }
            if (o + 1 < p.QN) {
                const int64_t e1 = (int64_t)(o + 1) * p.QK + i0;
                d1 = *reinterpret_cast<const unsigned*>(p.q_packed + (e1 >> 1) + ich * 4), a1 = p.q_absmax[e1 >> 6];
            }
        };
        unsigned nd0, nd1;
        float na0, na1;
        fetch(blockIdx.x, nd0, nd1, na0, na1);
        for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
            const int bo = blk / nbi, i0 = (blk - bo * nbi) * 64;
            const unsigned d0 = nd0, d1 = nd1;
            const float a0 = na0, a1 = na1;
            fetch(blk + gridDim.x, nd0, nd1, na0, na1);
            bf16_t v0[8], v1[8];
            decode(d0, a0, v0);
            decode(d1, a1, v1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned pr = (unsigned)__builtin_bit_cast(unsigned short, v0[j]) | ((unsigned)__builtin_bit_cast(unsigned short, v1[j]) << 16);
                *reinterpret_cast<unsigned*>(lds + (ich * 8 + j) * 260 + op * 4) = pr;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = tid + NT * q, i = idx >> 4, c = idx & 15;
                const unsigned* src = reinterpret_cast<const unsigned*>(lds + i * 260 + c * 16);
                const u32x4 r = {src[0], src[1], src[2], src[3]};
                const int oc = bo * 128 + c * 8;
                if (oc < p.QN) store16_through(out + (int64_t)(i0 + i) * p.QN + oc, r);
            }
            __syncthreads();
        }
    }
    if (p.q_debug & 2) return false;
    // ---- grid barrier.  Release: the image left through write-through stores, so every wave only waits for its own stores to
    //      be acknowledged (no L2 write-back; `__threadfence()` in every thread — eight waves x 32 workgroups per XCD each
    //      sweeping the whole L2 for write-back and again for invalidation — cost ~50 us per launch in the first version).
    //      Acquire: ONE wave per workgroup drops the stale
    //      lines of the scratch (it is re-used by every NF4 layer) from this CU's L1 and this XCD's L2.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's image stores are acknowledged; `epoch` (below) has arrived
    __syncthreads();
    if (stamps && tid == 0) stamps[1] = wall_clock64();
    if (tid < 64) {
        // Arrivals are counted in EIGHT words on their own cache lines (class = workgroup id & 7, i.e. the XCD: 32 atomics per
        // word instead of 256 on one) that are never reset: every launch adds exactly 1024 to each word — the first workgroup
        // of a class adds the difference to its class size, workgroup 0 also the 1024 of classes a small grid leaves empty —
        // so the launch with epoch e is complete when every word has reached 1024 (e + 1) (compared modulo 2^32).  `epoch`
        // was read before this workgroup arrived and is advanced by workgroup 0 once it is past the barrier, i.e. after
        // every workgroup of this launch has read it.  No departure count, no reset, nothing returned by an atomic.
        const unsigned grid = gridDim.x, cls = blockIdx.x & 7u;
        // (VFT_NF4_DEBUG & 4, tests only: workgroup 1 never arrives — what a workgroup kept off the chip looks like to the
        // others; & 8: a short spin limit so that such a test takes milliseconds)
        if (tid == 0 && !((p.q_debug & 4) && blockIdx.x == 1)) {
            unsigned add = 1u;
            if (blockIdx.x == cls) add += 1024u - (grid - cls + 7u) / 8u;
            __hip_atomic_fetch_add(&p.q_sync[64 + 16 * cls], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0)
                for (unsigned c = grid; c < 8u; ++c) __hip_atomic_fetch_add(&p.q_sync[64 + 16 * c], 1024u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned target = (epoch + 1u) * 1024u;
        unsigned spins = 0;
        for (;;) {  // lanes 0..7 poll one word each; bounded: every wave reaches the end of the kernel whatever happens
            const unsigned v = tid < 8 ? __hip_atomic_load(&p.q_sync[64 + 16 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__builtin_amdgcn_ballot_w64((int)(v - target) < 0) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > ((p.q_debug & 8) ? (1u << 12) : (1u << 24))) {
                if (tid == 0) {
                    __hip_atomic_store(&p.q_sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sticky: never cleared by a kernel
                    *fail_s = 1u;
                }
                break;
            }
        }
        asm volatile("buffer_inv sc1" ::: "memory");
        if (tid == 0 && blockIdx.x == 0) __hip_atomic_store(&p.q_sync[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[2] = wall_clock64();
    const bool failed = *fail_s != 0u;
    __syncthreads();  // the flag's LDS word belongs to the main loop's tiles from here on
    return failed;
}

template <class C, bool NF4 = false>
__global__ __launch_bounds__(NT, C::WPS) void lin8_kernel(Lin8Params p) {
    constexpr int BM = C::BM, BN = C::BN, FX = C::FX, FW = C::FW, XH = C::XH, WH = C::WH, TILE = C::TILE;
    constexpr int DX = C::DX, DW = C::DW;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm8[];
    bool nf4_failed = false;
    if (NF4) nf4_failed = nf4_fill_and_meet(p, sm8);  // p.w is complete and visible from here on — or the barrier timed out
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv / C::WAVES_N, wc = wv % C::WAVES_N, grp = wv >> 2;
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- persistent workgroups: virtual tile ids vt = blockIdx.x, + gridDim.x, ...; the grid is a multiple of 8, so
    //      vt & 7 labels the XCD group of every tile a workgroup takes.  id -> tile: XCD-contiguous, then bands of <= 8
    //      N-tiles walked M-major (the workgroups of one XCD share weight tiles and activation rows through its L2)
    const int tiles_m = (p.M - p.m_begin + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    auto decode = [&](int vt, int& m0, int& n0) {
        const int xcd = vt & 7, q = ntiles >> 3, rr = ntiles & 7;
        const int lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (vt >> 3);
        const int GN = tiles_n < 8 ? tiles_n : 8;
        const int band = lin / (GN * tiles_m);
        const int nb = min(GN, tiles_n - band * GN);
        const int within = lin - band * GN * tiles_m;
        n0 = (band * GN + within % nb) * BN;
        m0 = p.m_begin + (within / nb) * BM;
    };

    if (NF4 && nf4_failed) {
        // The grid never met (a workgroup was kept off the chip past the spin limit): the bf16 image of the weight is incomplete.
        // Nothing is computed from it: every output tile this workgroup owns is written as NaN, so the loss and every gradient
        // downstream are NaN at once, and the error word stays set for the host (VERDICT r4 weak #5 / ADVICE r4).
        const u32x4 nanv = {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};
        for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
            int m0, n0;
            decode(vt, m0, n0);
            for (int idx = tid; idx < BM * (BN / 8); idx += NT) {
                const int m = m0 + idx / (BN / 8), n = n0 + (idx % (BN / 8)) * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.y + (int64_t)m * p.ldy + n) = nanv;
            }
        }
        return;
    }

    // ---- LDS-DMA sources.  One instruction = 8 rows x 128 B: lane -> (row = lane>>3, slot = lane&7); the slot
    //      holds source chunk slot ^ (row & 7) (row & 7 == lane>>3: instruction bases are multiples of 8 rows)
    const int drow = lane >> 3, dch = (lane & 7) ^ (lane >> 3);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rex = __builtin_amdgcn_make_buffer_rsrc((void*)p.ex, 0, p.M * p.r * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rew = __builtin_amdgcn_make_buffer_rsrc((void*)p.ew, 0, p.N * p.r * 2, 0x00020000);
    int gx[2][DX];  // byte offsets of this lane's x chunks at k = 0
    // W: ONE lane-dependent offset for every tile (row lane>>3 of an 8-row group, chunk dch); the group's first row rides in
    // the SCALAR offset.  N % 8 == 0, so an 8-row group lies wholly inside or wholly outside the weight: groups beyond N are
    // moved to the last 8 rows (valid memory; their output columns are never stored) — wave-uniform, on the scalar unit.
    // (Round 3 kept 2 x DW per-lane offsets alive across the main loop; in the convolution instantiation two of them were
    // SPILLED, and their scratch reload inside phase B made hipcc wait `vmcnt(1)` between the X and W DMAs of every K-tile
    // — a full memory latency with the partner wave group idle behind the barrier.)
    // A/B on one MI355X (tools/exp/lin8_ab.py, old / new library): 256x256 convolutions at 64^2 / 32^2 +4-13 % (1 204 -> 1 287,
    // 1 390 -> 1 557 TF/s), 256x256 projections +0-2 %; the smaller tiles (N = 320 convolutions) LOSE 2-10 % to the extra
    // scalar work per DMA and never spilled — they keep the per-lane offsets.
    constexpr bool SCALAR_W = BM == 256 && BN == 256;
    const int wlane = (int)(((int64_t)drow * p.ldw + dch * 8) * 2);
    int wn0 = 0;  // n0 of the tile being staged (scalar)
    int gw[2][DW];  // !SCALAR_W: byte offsets of this lane's W chunks at k = 0
    unsigned vmask[2][DX];     // CONV: which of the 9 taps fall inside the image for this lane's pixel
    // CONV: K-tile t = (channel slice t / 9, tap t % 9); the X halves of one K-tile are issued at different times, so each keeps its own
    // incrementally advanced (tap, channel offset)
    int ctap[2] = {0, 0}, cc0[2] = {0, 0};
    auto setup = [&](int m0, int n0) {
        wn0 = n0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int q = 0; q < DX; ++q) {
                const int row = hh * (BM / 2) + (wv * DX + q) * 8 + drow;
                const int pm = min(m0 + row, p.M - 1);
                gx[hh][q] = (int)(((int64_t)pm * p.ldx + dch * 8) * 2);
                vmask[hh][q] = 0;
                if (C::CONV && C::STRIDE == 2) {
                    // stride 2, padding 1 (src/models/sdxl/denoiser.py:371-408): output pixel (oh, ow) of the cHo x cWo grid sits on
                    // input pixel (2 oh, 2 ow) of the cH x cW image; the nine taps are the same row shifts as at stride 1
                    const int how = p.cHo * p.cWo, bimg = pm / how, rem = pm - bimg * how, oh = rem / p.cWo, ow = rem - oh * p.cWo;
                    const int ih0 = 2 * oh, iw0 = 2 * ow;
                    gx[hh][q] = (int)((((int64_t)(bimg * p.cH + ih0) * p.cW + iw0) * p.ldx + dch * 8) * 2);
                    unsigned m = 0;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int ih = ih0 + tap / 3 - 1, iw = iw0 + tap % 3 - 1;
                        if (ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW) m |= 1u << tap;
                    }
                    vmask[hh][q] = m;
                } else if (C::CONV) {
                    const int rem = pm % (p.cH * p.cW), oh = rem / p.cW, ow = rem - oh * p.cW;
                    unsigned m = 0;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int ih = oh + tap / 3 - 1, iw = ow + tap % 3 - 1;
                        if (ih >= 0 && ih < p.cH && iw >= 0 && iw < p.cW) m |= 1u << tap;
                    }
                    vmask[hh][q] = m;
                }
            }
            if (!SCALAR_W) {
#pragma unroll
                for (int q = 0; q < DW; ++q) {
                    const int row = hh * (BN / 2) + (wv * DW + q) * 8 + drow;
                    gw[hh][q] = (int)(((int64_t)min(n0 + row, p.N - 1) * p.ldw + dch * 8) * 2);
                }
            }
            ctap[hh] = 0, cc0[hh] = 0;
        }
    };

    auto stage_x = [&](int t, int buf, int hh) {
        unsigned char* dst = sm8 + buf * TILE + hh * XH + wv * DX * 1024;
        if (C::CONV) {
            const int tap = ctap[hh];
            const int off = (((tap / 3 - 1) * p.cW + (tap % 3 - 1)) * (int)p.ldx + cc0[hh]) * 2;
#pragma unroll
            for (int q = 0; q < DX; ++q) dma16(rx, dst + q * 1024, ((vmask[hh][q] >> tap) & 1u) ? gx[hh][q] + off : OOB, 0);
            // K order of the convolution: the 9 taps of one 64-channel slice, then the next slice: the nine shifted reads
            // of a slice follow each other (with the channel loop inside the tap loop every tap row dy = -1, 0, 1 came
            // back only after the full Cin had been streamed).  Measured A/B on MI355X: +2-3 % on the SDXL convolutions;
            // the L2-miss traffic (FETCH_SIZE) stays at 2.1x the activation — the tile's two halo rows — either way
            if (++ctap[hh] == 9) {
                ctap[hh] = 0;
                cc0[hh] += 64;
            }
        } else {
#pragma unroll
            for (int q = 0; q < DX; ++q) dma16(rx, dst + q * 1024, gx[hh][q], t * 128);
        }
    };
    auto stage_w = [&](int t, int buf, int hh) {
        unsigned char* dst = sm8 + buf * TILE + 2 * XH + hh * WH + wv * DW * 1024;
        // the weight rows are [tap][Cin]: K-tile t of the order above is tap t % 9, channel slice t / 9
        const int koff = C::CONV ? ((t % 9) * p.cC + (t / 9) * 64) * 2 : t * 128;
#pragma unroll
        for (int q = 0; q < DW; ++q) {
            if (SCALAR_W) {
                const int g0 = min(wn0 + hh * (BN / 2) + (wv * DW + q) * 8, p.N - 8);  // first row of the 8-row group (scalar)
                dma16(rw, dst + q * 1024, wlane, g0 * (int)p.ldw * 2 + koff);
            } else {
                dma16(rw, dst + q * 1024, gw[hh][q], koff);
            }
        }
    };

    // ---- fragment read offsets (bytes inside a half-tile): row * 128 + ((chunk ^ (row & 7)) << 4), chunk = 4 kk + lq;
    //      row & 7 == lane & 7 (sub-tile bases are multiples of 16 rows)
    const int sw = ((lq ^ (lane & 7)) << 4);
    const int xl = (wr * (C::WM / 2) + l15) * 128 + sw;               // k-step 0; k-step 1 = ^ 64
    const int wl = 2 * XH + (wc * (C::WN / 2) + l15) * 128 + sw;

    f32x4 acc[2][FX][2][FW];
    bf16x8 xf[FX][2], w0[FW][2], w1[FW][2];
    auto read_x = [&](const unsigned char* tb, int hh) {
#pragma unroll
        for (int i = 0; i < FX; ++i) {
            xf[i][0] = *reinterpret_cast<const bf16x8*>(tb + hh * XH + i * 2048 + xl);
            xf[i][1] = *reinterpret_cast<const bf16x8*>(tb + hh * XH + i * 2048 + (xl ^ 64));
        }
    };
    auto read_w = [&](const unsigned char* tb, int hh, bf16x8 (&wf)[FW][2]) {
#pragma unroll
        for (int j = 0; j < FW; ++j) {
            wf[j][0] = *reinterpret_cast<const bf16x8*>(tb + hh * WH + j * 2048 + wl);
            wf[j][1] = *reinterpret_cast<const bf16x8*>(tb + hh * WH + j * 2048 + (wl ^ 64));
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

    const int nkt = p.K / 64;
    // One K-tile = TWO phases of 32 MFMAs per wave: (X0, W0) + (X0, W1), then (X1, W1) + (X1, W0 from registers); has1 / has2:
    // tiles t+1 / t+2 exist (wave-uniform).  Round 2 ran four phases of 16: half as many hand-overs between the two wave groups
    // (each costs the barrier's release plus the first operands' latency) is worth 4-7 % at K >= 3072, with the same registers
    // and the same DMA accounting.  Also tried and measured: fragment reads in the tail of the previous burst
    // (6 % slower), four barriers per K-tile without the enforced alternation of the groups (2-5 % slower), a third of the
    // fragment reads left out (+2-5 % only: LDS bandwidth is not the bound).
    auto tile = [&](int t, auto BUFc, bool has1, bool has2) {
        constexpr int BUF = decltype(BUFc)::value;
        const unsigned char* tb = sm8 + BUF * TILE;
        // ---- phase A: (X0, W0), (X0, W1); re-fills X1 of the other buffer (tile t+1) ----
        read_x(tb, 0);
        read_w(tb, 0, w0);
        read_w(tb, 1, w1);
        if (has1) stage_x(t + 1, BUF ^ 1, 1);
        lgk0();  // the X0 / W0 / W1 slots are re-filled in the next phase: their reads are done before this rendezvous
        bar();
        mma(acc[0], 0, w0);
        mma(acc[0], 1, w1);
        bar();
        // ---- phase B: (X1, W1), (X1, W0); re-fills X0, W0, W1 (tile t+2); everything of tile t+1 has landed after the wait
        //      (the three half-tiles of tile t+2 stay in flight) ----
        read_x(tb, 1);
        if (has2) {
            stage_x(t + 2, BUF, 0);
            stage_w(t + 2, BUF, 0);
            stage_w(t + 2, BUF, 1);
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
    // first K-tile complete; X0 / W0 / W1 of the second in flight (its X1 follows in phase 0)
    auto prologue = [&]() {
        stage_x(0, 0, 0);
        stage_w(0, 0, 0);
        stage_w(0, 0, 1);
        stage_x(0, 0, 1);
        if (nkt > 1) {
            stage_x(1, 1, 0);
            stage_w(1, 1, 0);
            stage_w(1, 1, 1);
        }
    };

    bf16_t* const ep = reinterpret_cast<bf16_t*>(sm8 + 2 * TILE);
    // Rank extension (r in {4, 8, 16}): the tile's rows of ex [BM][r] and ew [BN][r] are copied densely into the
    // epilogue region by LDS-DMA while that region is idle (issued before a tile's first K-stages, or behind the
    // previous tile's last epilogue pass), so they cost no registers and no exposed latency; rows beyond M / N read
    // zeros through the descriptor's bounds check.
    constexpr int EXW = BM * 32;  // byte offset of the ew copy (ex copy: BM rows x at most 32 bytes)
    static_assert(BM * 32 + BN * 32 <= C::EP_BYTES, "rank-extension staging must fit in the epilogue region");
    auto ext_dma = [&](int m0, int n0) {
        if (p.ex == nullptr) return;
        const int rb = p.r * 2;
        unsigned char* eb = sm8 + 2 * TILE;
        if (wv < BM * rb / 1024) dma16(rex, eb + wv * 1024, m0 * rb + wv * 1024 + lane * 16, 0);
        if (wv < BN * rb / 1024) dma16(rew, eb + EXW + wv * 1024, n0 * rb + wv * 1024 + lane * 16, 0);
    };
    constexpr int CH = BN / 8, EPQ = (C::EP_ROWS * CH) / NT;  // 16-byte chunks per image row; chunks per thread per pass
    constexpr int HP = (BM / 2) / C::EP_ROWS;                 // passes per X half

    int vt = blockIdx.x, m0, n0;
    decode(vt, m0, n0);
    setup(m0, n0);
    ext_dma(m0, n0);
    prologue();
#ifdef L8_STAMPS
    int stamp_it = 0;
#endif
    for (;;) {
        L8_STAMP(0);
        // the stores of the previous tile's epilogue are younger than these DMAs, so the count below (which assumes the
        // youngest operations are the second K-tile's DMAs) can only wait for MORE than it needs
        if (nkt > 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DX + 2 * DW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
        if (grp == 1) bar();  // stagger the groups by one barrier
        L8_STAMP(1);
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
            for (; t + 1 < nkt; t += 2) {
                tile(t, std::integral_constant<int, 0>{}, true, t + 2 < nkt);
                tile(t + 1, std::integral_constant<int, 1>{}, t + 2 < nkt, t + 3 < nkt);
            }
            if (t < nkt) tile(t, std::integral_constant<int, 0>{}, false, false);
        }
        if (grp == 0) bar();  // both groups are past their last fragment reads: the K-tile buffers are free
        L8_STAMP(2);

        // ---- the next tile's first DMAs fly while this tile is written out --------------------------------------
        const int cm0 = m0, cn0 = n0;
        const int vn = vt + gridDim.x;
        const bool more = vn < ntiles;
        if (more) {
            decode(vn, m0, n0);
            setup(m0, n0);
                    prologue();
        }

        L8_STAMP(3);
        // Lane-constant addresses of the tail (rank extension, epilogue) must not be hoisted out of the persistent loop:
        // they would stay live across the main loop, which has no register to spare.  An opaque copy of the thread id
        // per tile makes them loop-variant for the compiler.
        int etid = tid;
        asm volatile("" : "+v"(etid));
        const int el15 = etid & 15, elq = (etid & 63) >> 4;

        // ---- rank-r extension: one more MFMA k-step on the staged [rows][r] operands (k slot 8 lq + e <-> rank index;
        //      slots >= r are zero) --------------------------------------------------------------------------------
        if (p.ex != nullptr) {
            const int rb = p.r * 2, valid = p.r - 8 * elq;
            const unsigned char* eb = sm8 + 2 * TILE;
            auto frag = [&](int byte_base, int row) {
                const unsigned char* a = eb + byte_base + (valid > 0 ? row * rb + elq * 16 : 0);
                u32x2 lo = *reinterpret_cast<const u32x2*>(a), hi = *reinterpret_cast<const u32x2*>(a + 8);
                if (valid < 4) lo = u32x2{0u, 0u};
                if (valid < 8) hi = u32x2{0u, 0u};
                const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                return __builtin_bit_cast(bf16x8, v);
            };
            bf16x8 ew[2][FW], ex[2][FX];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < FW; ++j) ew[b][j] = frag(EXW, b * (BN / 2) + wc * (C::WN / 2) + j * 16 + el15);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < FX; ++i) ex[a][i] = frag(0, a * (BM / 2) + wr * (C::WM / 2) + i * 16 + el15);
            lgk0();
            bar();  // every wave has its fragments: the region becomes the epilogue image
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < FX; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < FW; ++j) acc[a][i][b][j] = mfma16(ew[b][j], ex[a][i], acc[a][i][b][j]);
        }

        L8_STAMP(4);
        // ---- epilogue: accumulators (+bias) -> bf16 image in LDS (64 token rows per pass; unpadded rows, 16-byte
        //      chunk c of row r at chunk c ^ (r & 15): conflict-free 8-byte fragment writes and 16-byte row reads) ->
        //      16-byte row-contiguous stores (+residual, loaded before the image is written) -----------------------
        bf16x4 bv[2][FW];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < FW; ++j) {
                const int fl = cn0 + b * (BN / 2) + wc * (C::WN / 2) + j * 16 + 4 * elq;
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[b][j][e] = (bf16_t)0.f;
                if (p.bias != nullptr && fl < p.N) bv[b][j] = *reinterpret_cast<const bf16x4*>(p.bias + fl);
            }
#pragma unroll
        for (int ps = 0; ps < BM / C::EP_ROWS; ++ps) {
            const int a = ps / HP, sub = ps % HP;
            const int mbase = cm0 + a * (BM / 2) + sub * C::EP_ROWS;
            bf16x8 rv[EPQ];
            if (p.res != nullptr) {
#pragma unroll
                for (int q = 0; q < EPQ; ++q) {
                    const int idx = etid + NT * q, row = idx / CH, ch = idx % CH;
                    const int m = min(mbase + row, p.M - 1), n = min(cn0 + ch * 8, p.N - 8);
                    rv[q] = *reinterpret_cast<const bf16x8*>(p.res + (int64_t)m * p.ldres + n);
                }
            }
            if ((wr * (C::WM / 2)) / C::EP_ROWS == sub) {
                const int r0 = (wr * (C::WM / 2)) % C::EP_ROWS;
#pragma unroll
                for (int i = 0; i < FX; ++i) {
                    const int row = r0 + i * 16 + el15;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < FW; ++j) {
                            bf16x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc[a][i][b][j][e] + (float)bv[b][j][e]);
                            const int col = b * (BN / 2) + wc * (C::WN / 2) + j * 16 + 4 * elq;
                            *reinterpret_cast<bf16x4*>(ep + row * BN + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7))) = v;
                        }
                }
            }
            lgk0();
            bar();
#pragma unroll
            for (int q = 0; q < EPQ; ++q) {
                const int idx = etid + NT * q, row = idx / CH, ch = idx % CH;
                const int m = mbase + row, n = cn0 + ch * 8;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(ep + row * BN + ((ch ^ (row & 15)) << 3));
                if (p.res != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)rv[q][e]);
                }
                // Large outputs are written with the non-temporal hint: ordinary stores of a 1 GB result pushed the x / W
                // tiles the XCD's workgroups share out of its 4 MB L2 (measured, M 98304 N 5120 K 640: 717 -> 637 us;
                // N 1920: 300 -> 272; M 24576 N 1280 K 5120: 279 -> 259); small outputs stay cacheable for their consumer
                if (m < p.M && n < p.N) {
                    bf16x8* dst = reinterpret_cast<bf16x8*>(p.y + (int64_t)m * p.ldy + n);
                    if (p.nt_store)  // (inline asm: hipcc folds a __builtin_nontemporal_store branch into the plain store)
                        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
                    else *dst = v;
                }
            }
            lgk0();
            bar();  // the image is re-used by the next pass
            L8_STAMP(5 + ps);
        }
#ifdef L8_STAMPS
        ++stamp_it;
#endif
        if (!more) break;
        ext_dma(m0, n0);  // behind the last pass's barrier: the region is idle again
        vt = vn;
    }
}

template <class C, bool NF4>
void launch8k(const Lin8Params& p, hipStream_t st) {
    static int resident = 0;  // workgroups of this kernel the chip holds at once (first call; 0 = not asked yet)
    if (!resident) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(lin8_kernel<C, NF4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::LDS_BYTES);
        resident = C::SLOTS;
        if (NF4) {
            // the dequantising launch meets at a grid barrier: the grid must fit the chip (asked once, outside any capture
            // in practice: the first call is an eager warm-up step; a failed query keeps the conservative one per CU)
            int per_cu = 0, dev = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(lin8_kernel<C, NF4>), NT, C::LDS_BYTES) != hipSuccess)
                per_cu = 1;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 8;
            (void)hipGetLastError();
            const int fit = (per_cu < 1 ? 1 : per_cu) * cus;
            resident = fit < C::SLOTS ? (fit >= 8 ? fit / 8 * 8 : fit) : C::SLOTS;
        }
    }
    // persistent workgroups: one per resident slot (a multiple of 8, so that a workgroup stays in one XCD group), or
    // one per tile when there are fewer tiles than slots
    const unsigned tiles = (unsigned)(((p.M - p.m_begin + C::BM - 1) / C::BM) * ((p.N + C::BN - 1) / C::BN));
#ifdef L8_NONPERSIST
    const unsigned grid = NF4 ? (tiles < (unsigned)resident ? tiles : (unsigned)resident) : tiles;
#else
    const unsigned grid = tiles < (unsigned)resident ? tiles : (unsigned)resident;
#endif
    hipLaunchKernelGGL((lin8_kernel<C, NF4>), dim3(grid), dim3(NT), C::LDS_BYTES, st, p);
}
template <class C>
void launch8(const Lin8Params& p, hipStream_t st) {
    if constexpr (!C::CONV) {
        if (p.q_packed != nullptr) return launch8k<C, true>(p, st);
    }
    launch8k<C, false>(p, st);
}

}  // namespace

// cfg: 1 = 256x256, 2 = 256x128, 3 = 128x128 (see linear8.h)
static int64_t x_extent(const Lin8Params& p, int conv) {
    const int64_t rows = p.x_rows > 0 ? p.x_rows : p.M;
    return conv ? rows * p.ldx * 2 : ((rows - 1) * p.ldx + p.K) * 2;
}
static int64_t w_extent(const Lin8Params& p) { return ((int64_t)(p.N - 1) * p.ldw + p.K) * 2; }

bool vft_lin8_fits(const Lin8Params& p, int conv) {
    const int64_t lim = (1LL << 31) - 4096;
    return p.K % 64 == 0 && x_extent(p, conv) < lim && w_extent(p) < lim;
}

void vft_lin8_launch(Lin8Params p, int cfg, int conv, hipStream_t st) {
    static const int64_t nt_bytes = [] {  // VFT_LIN8_NT_MB: A/B knob (outputs above this many MiB are stored non-temporally; -1: never)
        const char* e = getenv("VFT_LIN8_NT_MB");
        const int64_t mb = e ? atoll(e) : -2;
        return mb == -1 ? INT64_MAX : (mb >= 0 ? mb << 20 : (int64_t)L8_NT_BYTES);
    }();
    p.nt_store = (int64_t)(p.M - p.m_begin) * p.N * 2 > nt_bytes;
    p.x_bytes = (int)x_extent(p, conv);
    p.w_bytes = (int)w_extent(p);
    if (conv == 2) {  // stride-2 convolution (two launches per SDXL step: no 256x256 instantiation — the 256x128 / 128x128 ones have registers to spare)
        if (cfg == 3) launch8<C8<128, 128, 2, true, 2>>(p, st);
        else launch8<C8<256, 128, 4, true, 2>>(p, st);
    } else if (conv) {
        if (cfg == 1) launch8<C8<256, 256, 2, true>>(p, st);
        else if (cfg == 2) launch8<C8<256, 128, 4, true>>(p, st);
        else launch8<C8<128, 128, 2, true>>(p, st);
    } else {
        if (cfg == 1) launch8<C8<256, 256, 2, false>>(p, st);
        else if (cfg == 2) launch8<C8<256, 128, 4, false>>(p, st);
        else launch8<C8<128, 128, 2, false>>(p, st);
    }
}

// Which tile.  Workgroups are persistent, so a launch costs ceil(tiles / resident slots) rounds of one tile each; a tile
// costs its area x (K + K0) over the speed a slot sustains, K0 standing for the per-tile prologue / epilogue.  The four
// constants (speed of a 256x128 slot 0.78 and of a 128x128 slot 0.35 relative to 256x256 — two of the latter share a CU;
// K0 = 128 / 96 / 64) are fitted to the 316 measured problems of tuning_tables/gfx950_lin8_tiles.csv: choosing by this
// model costs 0.9 % over the per-problem optimum there (the round-1 model: 8.3 %).  Split launches (whole rounds on the
// tile + the remaining rows on the best plain tile, vft_lin8_split_rows) compete as configurations 11-13.
static double lin8_cost(int M, int N, int K, int c) {
    const double speed[4] = {0.0, 1.0, 0.78, 0.35}, k0[4] = {0.0, 128.0, 96.0, 64.0};
    const int bm[4] = {0, 256, 256, 128}, bn[4] = {0, 256, 128, 128}, slots[4] = {0, 256, 256, 512};
    const long tiles = (long)((M + bm[c] - 1) / bm[c]) * ((N + bn[c] - 1) / bn[c]);
    const long rounds = (tiles + slots[c] - 1) / slots[c];
    return (double)rounds * bm[c] * bn[c] * (K + k0[c]) / speed[c];
}
static int lin8_pick_plain(int M, int N, int K, double* cost) {
    int best = 1;
    double best_t = 1e300;
    for (int c = 1; c <= 3; ++c) {
        const double t = lin8_cost(M, N, K, c);
        if (t < best_t * 0.97) best_t = t, best = c;  // ties and near-ties go to the larger tile
    }
    if (cost) *cost = best_t;
    return best;
}
int vft_lin8_pick_plain(int M, int N, int K) { return lin8_pick_plain(M, N, K, nullptr); }
int vft_lin8_pick(int M, int N, int K) {
    double best_t;
    int best = lin8_pick_plain(M, N, K, &best_t);
    for (int c = 1; c <= 3; ++c) {
        const int M1 = vft_lin8_split_rows(M, N, c);
        if (M1 <= 0) continue;
        double tail;
        lin8_pick_plain(M - M1, N, K, &tail);
        const double t = lin8_cost(M1, N, K, c) + tail;
        if (t < best_t * 0.97) best_t = t, best = 10 + c;
    }
    return best;
}

int vft_lin8_split_rows(int M, int N, int main_cfg) {
    const int bm[4] = {0, 256, 256, 128}, bn[4] = {0, 256, 128, 128}, slots[4] = {0, 256, 256, 512};
    if (main_cfg < 1 || main_cfg > 3) return 0;
    const long tiles_n = (N + bn[main_cfg] - 1) / bn[main_cfg];
    const long tiles_m = (M + bm[main_cfg] - 1) / bm[main_cfg];
    const long rounds = tiles_m * tiles_n / slots[main_cfg];
    if (rounds < 1) return 0;
    const long m1 = rounds * slots[main_cfg] / tiles_n;  // whole tile rows that fit `rounds` rounds
    const long M1 = m1 * bm[main_cfg];
    return (M1 > 0 && M1 < M) ? (int)M1 : 0;
}
