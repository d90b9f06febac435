// Internal interface between attn.hip and wgrad8.hip: the batched form of the weight-gradient GEMM.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// out[bi] (bf16 [R][ldo]) = scale * A[bi]^T B[bi] for bi = bo * inner + bin:
//   A[bi] = a + bi * a_bstride                         [M][lda]   (reduction index on the rows)
//   B[bi] = b + bo * b_outer + (bin / G) * b_inner     [M][ldb]
//   out[bi] = out + bo * o_outer + bin * o_inner
// Used by the attention backward at head_dim >= 128: dQ[q][d] = scale * sum_key dS^T[key][q] K[key][d] per (batch, query head),
// G = query heads per kv head.  scratch: vft_wg8_batched_scratch_floats(...) fp32 partial tiles (ordered sum, reproducible).
struct Wg8Batch {
    int nbatch, inner, G;
    int64_t a_bstride, b_outer, b_inner, o_outer, o_inner;
    float scale;
};
int64_t vft_wg8_batched_scratch_floats(int64_t M, int R, int Cc, int nbatch);
int vft_wg8_batched(const void* a, const void* b, void* out, float* scratch, int64_t M, int R, int Cc, int64_t lda, int64_t ldb,
                    int64_t ldo, const Wg8Batch& bt, hipStream_t stream);
