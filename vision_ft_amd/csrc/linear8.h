// Internal interface between linear.hip (C-ABI entry points, argument checks, kernel choice) and linear8.hip
// (the second-generation main loop).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vft_common.h"

struct Lin8Params {
    const bf16_t* x;     // [M][ldx]   (CONV: [B*H*W][ldx = Cin], gathered per tap)
    const bf16_t* w;     // [N][ldw]
    bf16_t* y;           // [M][ldy]
    const bf16_t* bias;  // [N] or null
    const bf16_t* res;   // [M][ldres] or null
    const bf16_t* ex;    // [M][r] or null
    const bf16_t* ew;    // [N][r]
    int M, N, K, r;
    int x_rows;          // rows of x the descriptor may address (0: M) — a split launch's first part still reads the
                         // convolution halo rows beyond its own M
    int m_begin;         // first row this launch covers (rows [m_begin, M)); 0 except for the tail launch of a split problem
    int64_t ldx, ldw, ldy, ldres;
    int cH, cW, cC;      // CONV: image height / width / input channels (K = 9 * cC, cC % 64 == 0)
    int cHo, cWo;        // CONV, stride 2: the output grid (cH / 2 x cW / 2); M = B * cHo * cWo, x_rows = B * cH * cW
    int x_bytes, w_bytes;  // extents of x and w for the buffer descriptors (set by vft_lin8_launch; < 2^31)
    int nt_store;          // y is stored with the non-temporal hint (set by vft_lin8_launch: outputs above 16 MiB)
    // NF4 base weight (vft_linear_nf4 with a workspace): q_packed != null makes the launch DEQUANTISE the weight first — every
    // workgroup converts its share of the packed codes into `w` (then a bf16 scratch [N][ldw = K] the caller provides), the
    // grid meets at a barrier (q_sync), and the main loop stages that scratch like any bf16 weight.  One dequantisation per
    // GEMM instead of one per row tile; the scratch (<= 28 MB for Lumina2) lives in the Infinity Cache, the 4-bit codes stay
    // the only copy of the weight that persists in HBM.  Logical weight [QN][QK] (out, in), blocksize 64 along in.
    const uint8_t* q_packed;  // [QN*QK/2], element 2i in the high nibble
    const float* q_absmax;    // [QN*QK/64]
    const float* q_code;      // [16]
    int QN, QK;
    int q_trans;              // 0: w[n][k] = W[n][k] (forward, N = QN, K = QK);  1: w[n][k] = W[k][n] (data gradient, N = QK, K = QN)
    unsigned* q_sync;         // header of the workspace, zero before the first launch: [0] launch counter, [2] error flag,
                              // [64 + 16 c] arrival word of workgroup class c (8 classes); [128 ...] cycle stamps (experiments)
    int q_debug;              // timing experiments only (VFT_NF4_DEBUG): 1 = skip the conversion, 2 = skip the barrier + fences (WRONG results)
};

// true when the operands fit the 32-bit offsets of the buffer descriptors (otherwise: first-generation kernel)
bool vft_lin8_fits(const Lin8Params& p, int conv);  // conv: 0 = linear, 1 = 3x3 stride 1, 2 = 3x3 stride 2

// cfg: 1 = 256x256 tile (128 KiB LDS), 2 = 256x128 (96 KiB), 3 = 128x128 (64 KiB, two workgroups per CU).  K % 64 == 0.
void vft_lin8_launch(Lin8Params p, int cfg, int conv, hipStream_t st);
// tile configuration (1-3, or 11-13 = split launch) with the least modelled time for an (M, N, K) problem on 256 CUs;
// _plain: among the one-launch configurations 1-3 only (the tail of a split launch)
int vft_lin8_pick(int M, int N, int K);
int vft_lin8_pick_plain(int M, int N, int K);
// Split launch (cfg 11 / 12 / 13 = main tile 1 / 2 / 3): the persistent workgroups take whole rounds of tiles only — rows
// [0, M1) with M1 the largest tile-row count whose tiles fill an integer number of rounds — and the remaining rows run
// as a second launch on the tile vft_lin8_pick chooses for them (smaller tiles: the ragged last round of a big-tile
// launch, e.g. 828 tiles on 256 slots = 3.23 rounds, costs a whole round otherwise).  Returns M1 (0: do not split).
int vft_lin8_split_rows(int M, int N, int main_cfg);
