// NF4 -> bf16 weight materialisation:  out[QN][QK] = bf16(code[c] * absmax) in the bitsandbytes layout
// (src/modules/quant/bnb.py:37-129 -> bitsandbytes 0.48.2 kDequantizeBlockwise: packed uint8, element 2i in the high
// nibble, blocksize 64 along the flattened index, absmax fp32 per block or double-quantised: code2[q8] * absmax2[blk / 256]
// + offset).  Used by the optional bf16 compute cache (modules/quant/nf4.py), the fallback for shapes the dequantising GEMM
// does not take, tests and tooling; the training path dequantises inside vft_linear_nf4 (linear8.hip: nf4_fill_and_meet).
// (Round 1's 128x128 register-staged GEMM with an NF4 staging mode lived here as gemm.hip; it was superseded by
// linear.hip / linear8.hip in round 2 and removed in round 5.)
#include "vft_common.h"

namespace {

struct DequantParams {
    const uint8_t* q_packed;    // QN*QK/2 bytes
    const uint8_t* q_absmax8;   // [nblocks] (double quant) or null
    const float* q_absmax32;    // [nblocks] fp32 absmax (no double quant) or null
    const float* q_absmax2;     // [ceil(nblocks/256)]
    const float* q_code2;       // [256]
    const float* q_code;        // [16]
    float q_offset;
    int QN, QK;
};

__device__ __forceinline__ float nf4_absmax(const DequantParams& p, int64_t blk) {
    if (p.q_absmax8) return p.q_code2[p.q_absmax8[blk]] * p.q_absmax2[blk >> 8] + p.q_offset;
    return p.q_absmax32[blk];
}

__global__ __launch_bounds__(256) void nf4_dequant_kernel(DequantParams p, bf16_t* __restrict__ out) {
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

}  // namespace

extern "C" int vft_nf4_dequant(void* out, const void* packed, const void* absmax, int absmax_is_u8,
                               const float* nested_absmax, const float* nested_code, const float* code,
                               float nested_offset, int out_features, int in_features, void* stream) {
    VFT_CHECK_ARG(out != nullptr && ((uintptr_t)out & 15) == 0, "vft_nf4_dequant: bad output pointer");
    VFT_CHECK_ARG(packed && absmax && code, "vft_nf4_dequant: null quantisation state");
    VFT_CHECK_ARG(!absmax_is_u8 || (nested_absmax && nested_code), "vft_nf4_dequant: double-quantised absmax needs nested_absmax and nested_quant_map");
    VFT_CHECK_ARG(out_features > 0 && in_features > 0 && in_features % 64 == 0 && ((uintptr_t)packed & 15) == 0,
                  "vft_nf4_dequant: weight must have in_features %% 64 == 0 and a 16-byte aligned payload");
    DequantParams p{};
    p.q_packed = (const uint8_t*)packed;
    p.q_absmax8 = absmax_is_u8 ? (const uint8_t*)absmax : nullptr;
    p.q_absmax32 = absmax_is_u8 ? nullptr : (const float*)absmax;
    p.q_absmax2 = nested_absmax, p.q_code2 = nested_code, p.q_code = code, p.q_offset = nested_offset;
    p.QN = out_features, p.QK = in_features;
    const int64_t nvec = (int64_t)out_features * in_features / 8;
    int64_t nb = (nvec + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(nf4_dequant_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, (bf16_t*)out);
    VFT_CHECK_LAUNCH("vft_nf4_dequant");
    return 0;
}
