/*
 * vft_hip.h — C ABI of libvft_hip.so: the MI355X (gfx950) kernels behind vision-ft's denoiser
 * forward/backward hot path.
 *
 * The reference (p1atdev/vision-ft) has no FFI of its own: its operator API is Python call
 * signatures.  Each entry point below names the reference call site whose
 * third-party CUDA kernel it replaces; the Python host mirror in vision_ft_amd/ keeps the
 * reference signatures and calls these through ctypes (vision_ft_amd/_C.py).
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes only; every buffer is owned by the caller (PyTorch) and merely
 *     borrowed for the duration of the enqueue.  The library keeps no state besides a
 *     thread-local error string.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises.
 *   - return 0 on success; non-zero on a rejected argument (1) or a failed launch (2), with the
 *     reason available from vft_last_error().  Never aborts, never exits.
 *   - bf16 tensors are passed as `const void*` to raw bf16 storage; fp32 as float*.
 *   - re-entrant; safe under torch.utils.checkpoint recompute; graph-capturable (no allocation,
 *     no host synchronisation inside).
 */
#ifndef VFT_HIP_H
#define VFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- housekeeping ------------------------------------------------------------------------ */
int vft_version(void);
const char* vft_last_error(void);
/* gcnArchName of `device` into name[cap]; returns the CU count or -1. */
int vft_device_info(int device, char* name, int cap);

/* ---- attention ---------------------------------------------------------------------------
 * Replaces flash_attn_func / xformers.memory_efficient_attention / F.scaled_dot_product_attention
 * as dispatched by   src/modules/attention.py:98-159  (scaled_dot_product_attention)
 *              and   src/modules/attention.py:31-95   (scaled_qkv_attention, AuraFlow).
 * softmax(scale * Q K^T [+ key-padding mask]) V, non-causal, dropout 0.
 *   q: [B,Hq,Sq,D]  k,v: [B,Hkv,Sk,D]  out: [B,Hq,Sq,D]   (bf16; logical shapes — each tensor is
 *   addressed through strides[3] = {batch, head, seq} in elements, head_dim contiguous, so both
 *   (B,H,S,D) and (B,S,H,D) memory layouts are accepted without a copy.  Strides must be
 *   multiples of 8 elements and base pointers 16-byte aligned.)
 *   lse: [B,Hq,Sq] fp32 natural-log log-sum-exp of the scaled scores (saved for backward).
 *   kv_len: optional [B] int32 — keys >= kv_len[b] are masked (Lumina2's prefix bool mask,
 *   src/models/lumina2/denoiser.py:150-155); NULL = no mask.
 *   GQA: Hq % Hkv == 0, query head h reads kv head h / (Hq/Hkv)  (the reference materialises the
 *   repeat, src/models/lumina2/denoiser.py:146-147).
 *   D in {64, 96, 128, 256}.
 */
int vft_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                 const int32_t* kv_len, int B, int Hq, int Hkv, int Sq, int Sk, int D,
                 const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                 const int64_t* o_strides, float scale, void* stream);

/* Backward of the above.  delta_ws: [B,Hq,Sq] fp32 scratch.  dk/dv are per kv head (the sum over
 * the query heads of a GQA group happens in-kernel).  Deterministic (no atomics). */
int vft_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                 const float* lse, void* dq, void* dk, void* dv, float* delta_ws,
                 const int32_t* kv_len, int B, int Hq, int Hkv, int Sq, int Sk, int D,
                 const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                 const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides,
                 const int64_t* dk_strides, const int64_t* dv_strides, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFT_HIP_H */
