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

/* Backward of the above.  ws: fp32 scratch of ws_floats elements, at least B*Hq*Sq (delta = rowsum(dO*O));
 * allocate vft_attn_bwd_workspace(...) floats to let (a) head_dim 64 with Sk > 128 run the ONE-PASS backward (dQ, dK
 * and dV from a single sweep: five matrix products per block instead of seven; the workgroup that walks a (batch, head)
 * pair's key blocks keeps the running dQ in an fp32 plane of the scratch — B*Hq*ceil(Sq/64)*64*D floats per key-range
 * slice), (b) head_dim 128 / 256 without a key mask (AuraFlow; Sq a multiple of 8, at least 512 queries and keys) run
 * the STORED-dS backward: the dK/dV kernel writes dS^T (bf16, B*Hq*Sk*roundup(Sq,32) elements of the scratch) and dQ is one
 * batched GEMM on the weight-gradient kernel — again five products instead of seven — and (c) short key sequences
 * (cross-attention, Sk = 77: sdxl/denoiser.py:224-262) split their query tiles over more workgroups — the partial dK/dV
 * are then summed in a fixed order.  With less scratch the dq + dkdv kernel pair runs.  dk/dv are per kv head (the sum over the query heads of a GQA group happens in-kernel).  Deterministic (no
 * atomics, fixed summation order). */
int64_t vft_attn_bwd_workspace(int B, int Hq, int Hkv, int Sq, int Sk, int D);
int vft_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                 const float* lse, void* dq, void* dk, void* dv, float* ws,
                 const int32_t* kv_len, int B, int Hq, int Hkv, int Sq, int Sk, int D,
                 const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                 const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides,
                 const int64_t* dk_strides, const int64_t* dv_strides, float scale, int64_t ws_floats,
                 void* stream);
/* Tuning / A-B control of the attention kernels (process-wide; the trainers never call it): 0 = automatic, 1 = the
 * first-generation 4-wave kernels for every head_dim, 2 = the 8-wave dK/dV kernel for head_dim >= 128, 3 = the dq + dkdv
 * kernel pair where the one-pass / stored-dS backward would run. */
int vft_attn_variant(int variant);

/* ---- LoRA -----------------------------------------------------------------------------------
 * Replace the low-rank branch of LoRALinear.forward, src/modules/peft/lora.py:92-104
 *   linear(x) + lora_up(lora_down(dropout(x))) * (alpha / rank)
 * and the skinny GEMMs autograd derives from it.  All tensors bf16, row-major 2-D views (tokens
 * flattened to M rows); ld* are row strides in elements; lora_down is (r,K), lora_up is (N,r), both
 * contiguous; r <= 16; K, N multiples of 8.
 *
 * vft_lora_fwd:     y[M,N] += scale * (x[M,K] . lora_down^T) . lora_up^T, in place on the base GEMM's
 *                   output.  Also writes t_out[M,r] = x . lora_down^T and up_t_out[r,N] = lora_up^T
 *                   (both saved for backward; either may be NULL).
 * vft_lora_bwd_dx:  dx[M,K] += scale * (dy[M,N] . lora_up) . lora_down, in place on the base dgrad;
 *                   takes lora_up^T (r,N) from the forward; writes u_out[M,r] = dy . lora_up.
 * vft_lora_wgrad2:  d_down[r,K] = scale * u^T x ;  d_up[N,r] = scale * dy^T t   (one launch pair,
 *                   deterministic two-stage reduction; workspace >= vft_lora_wgrad_workspace()).
 */
/* residual (optional, [M][ldres]): the transformer block's `x + layer(...)` add (sdxl/denoiser.py:273-284) folded
 * into the same pass: y = bf16(y + lora) + residual. */
int vft_lora_fwd(const void* x, void* y, const void* lora_down, const void* lora_up, void* t_out,
                 void* up_t_out, const void* residual, int M, int K, int N, int r, int64_t ldx,
                 int64_t ldy, int64_t ldres, float scale, void* stream);
/* t[M][r] = scale * (x . lora_down^T) only: the adapter's low-rank activations, for consumers that apply the update
 * themselves (the rank extension of vft_linear_fwd / vft_linear_nf4); scale = alpha / rank lets the
 * consumer use lora_up as stored (peft/lora.py:92-104), scale = 1 gives the plain activations. */
int vft_lora_down(const void* x, const void* lora_down, void* t_out, int M, int K, int r, int64_t ldx,
                  float scale, void* stream);
int vft_lora_wgrad_target(int workgroups); /* tuning / A-B: workgroups the weight-gradient row blocking aims at; 0 = automatic */
int vft_lora_down_variant(int variant); /* tuning / A-B: 0 = automatic (32 token rows per workgroup, 64 for rank > 8 at M >= 16384), 1 = 16, 2 = 64, 3 = 32 */
int vft_lora_bwd_dx(const void* dy, void* dx, const void* lora_up_t, const void* lora_down, void* u_out,
                    int M, int N, int K, int r, int64_t lddy, int64_t lddx, float scale, void* stream);
size_t vft_lora_wgrad_workspace(int M, int K, int N, int r);
int vft_lora_wgrad2(const void* x, const void* u, const void* dy, const void* t, void* d_down,
                    void* d_up, float* workspace, size_t ws_bytes, int M, int K, int N, int r,
                    int64_t ldx, int64_t lddy, float scale, void* stream);
/* Grouped / in-place form of vft_lora_wgrad2.  `groups` (1..3) adapters share the input x — SDXL's fused q|k|v
 * projection (sdxl/denoiser.py:57-64): u (M,r) and t (M,r) hold the groups' rank-`rsub` columns side by side (zero
 * padded to r), dy (M,N) their N/groups output columns side by side.  The gradients go straight into the caller's dense
 * tensors d_down[g] (rsub,K) and d_up[g] (N/groups,rsub) — with accumulate != 0 they are ADDED to the contents, which
 * is what torch's AccumulateGrad would do with a returned gradient (the trainer points param.grad into one flat buffer,
 * src/trainer/common.py:355-365 accumulation semantics). */
int vft_lora_wgrad2g(const void* x, const void* u, const void* dy, const void* t, void* const* d_down,
                     void* const* d_up, int groups, int rsub, int accumulate, float* workspace, size_t ws_bytes,
                     int M, int K, int N, int r, int64_t ldx, int64_t lddy, float scale, void* stream);

/* Deferred reductions: vft_lora_wgrad2g_defer computes the partial sums only and records its two reductions as entries
 * `slot`, `slot + 1` of a device table (vft_lora_wgrad_desc_bytes() bytes per entry; written by the kernel, so no
 * host-to-device copy and capturable); vft_lora_wgrad_reduce_many(table, n, max_elements) then runs all n queued
 * reductions in one launch (max_elements = the largest K*r or N*r among them).  The workspaces must stay untouched in
 * between.  One training step of SDXL-LoRA: 560 + 560 launches become 560 + 1. */
size_t vft_lora_wgrad_desc_bytes(void);
int vft_lora_wgrad2g_defer(const void* x, const void* u, const void* dy, const void* t, void* const* d_down,
                           void* const* d_up, int groups, int rsub, int accumulate, float* workspace,
                           size_t ws_bytes, int M, int K, int N, int r, int64_t ldx, int64_t lddy, float scale,
                           void* table, int slot, void* stream);
int vft_lora_wgrad_reduce_many(const void* table, int n, int64_t max_elements, void* stream);

/* Round 5 — the backward pass of LoRALinear (peft/lora.py:92-104 differentiated) with dy and x read ONCE each for the
 * rank-sized work:
 *   vft_lora_bwd_dy: the pass over dy[M,N] yields BOTH d(lora_up) = scale_w * dy^T t (partial sums + reduction exactly as
 *                    vft_lora_wgrad2g / _defer do for that half; table == NULL: reduced at once) AND
 *                    u_out[M,r] (bf16) = scale_u * dy . lora_up — the operand the data gradient's rank extension needs
 *                    (vft_linear_fwd on the transposed weight, ext_x = u) — through per-column-tile partial sums in u_part
 *                    (vft_lora_u_part_floats(M, N, r) floats) added in tile order (deterministic).  up_t = lora_up^T (r,N),
 *                    r the padded rank (4 / 8 / 16).  Replaces vft_lora_down(dy, up_t): one full read of dy per layer.
 *   vft_lora_bwd_x:  d(lora_down) = scale_w * u^T x, the other half of vft_lora_wgrad2g.
 * Workspaces: vft_lora_bwd_workspace(M, N or K, r) bytes each; groups / rsub / accumulate / table / slot as above (ONE
 * table entry per call). */
int64_t vft_lora_u_part_floats(int M, int N, int r);
size_t vft_lora_bwd_workspace(int M, int cols, int r, int with_u); /* with_u: 1 for the dy pass, 0 for the x pass */
int vft_lora_bwd_dy(const void* dy, const void* t, const void* up_t, void* u_out, float* u_part, void* const* d_up,
                    int groups, int rsub, int accumulate, float* workspace, size_t ws_bytes, int M, int N, int r,
                    int64_t lddy, float scale_u, float scale_w, void* table, int slot, void* stream);
int vft_lora_bwd_x(const void* x, const void* u, void* const* d_down, int groups, int rsub, int accumulate,
                   float* workspace, size_t ws_bytes, int M, int K, int r, int64_t ldx, float scale_w, void* table,
                   int slot, void* stream);

/* All derived adapter operands of a model in one launch.  `table` (device memory) holds n descriptors of seven int64:
 * {src0, src1, src2, dst, kind | groups << 32, r | rp << 32, L}; kind 0: dst (rp,L) = the groups' lora_down (r,L) stacked
 * and zero-padded to rp rows; kind 1: its transpose (L,rp); kind 2: dst (groups*L, rp) = blockdiag of the groups'
 * lora_up (L,r), zero-padded; kind 3: its transpose.  These are the forms the rank extension of vft_linear_fwd and
 * vft_lora_down consume (a fused q|k|v projection uses groups = 3); the reference needs none of them because it runs
 * the adapter as two extra GEMMs (peft/lora.py:92-104). */
int vft_lora_pack(const void* table, int n, void* stream);

/* ---- row norms -----------------------------------------------------------------------------
 * LayerNorm / RMSNorm over the last dimension with fp32 statistics, optional per-channel affine
 * (weight, bias: bf16 [C]) and optional AdaLN modulation y = norm(x)*(1+scale[b]) + shift[b]
 * (scale/shift: bf16 [rows/rows_per_batch][C]).  kind: 0 = LayerNorm, 1 = RMSNorm.
 * Replaces nn.LayerNorm (sdxl/denoiser.py:250-252), Fp32LayerNorm + modulate
 * (auraflow/denoiser.py:59-76), nn.RMSNorm + (1+scale) (lumina2/denoiser.py:280-313),
 * FP32LayerNorm (modules/norm.py:9-17).  mean/rstd: fp32 [rows], saved for backward.
 */
int vft_norm_fwd(const void* x, void* y, const void* weight, const void* bias, const void* scale,
                 const void* shift, float* mean, float* rstd, int64_t rows, int C, int64_t ldx,
                 int rows_per_batch, int kind, float eps, void* stream);
/* dres, dres2 (optional, [rows][C]): gradients arriving through the residual path(s) around the norm — the block input
 * feeds the norm and one or two skip connections (auraflow/denoiser.py:392-422: both gated adds of a block take the
 * block input); dx = norm_bwd + dres + dres2 */
int vft_norm_bwd_dx(const void* dy, const void* x, const void* weight, const void* scale,
                    const float* mean, const float* rstd, void* dx, const void* dres, const void* dres2, int64_t rows,
                    int C, int64_t ldx, int rows_per_batch, int kind, void* stream);

/* ---- weight gradient of a trainable projection ----------------------------------------------
 * dw[R][Cc] (+)= dy[M][R]^T . x[M][Cc]   (bf16 operands, fp32 accumulation): what autograd runs as torch.mm for the
 * weight of F.linear when the base weights train (full fine-tune; src/models/sdxl/denoiser.py:62-64,135-137,183-204,
 * src/modules/peft/lora.py:92-104 with the adapters merged away).  The reduction over the M tokens is split over row
 * slices; partial tiles go to `scratch` (vft_linear_wgrad_scratch_floats(M, R, Cc) floats) and are added in slice
 * order: no atomics, bit-reproducible.  accumulate != 0: dw += ...; out_f32: dw is fp32 instead of bf16.
 * R % 8 == 0, Cc % 8 == 0, strides in elements (lddy, ldx % 8 == 0; lddw % 4 == 0), operands below 2 GiB. */
int64_t vft_linear_wgrad_scratch_floats(int64_t M, int R, int Cc);
int vft_linear_wgrad(const void* dy, const void* x, void* dw, float* scratch, int64_t M, int R, int Cc,
                     int64_t lddy, int64_t ldx, int64_t lddw, int accumulate, int out_f32, void* stream);

/* Weight gradient of the 3x3 / stride 1 / padding 1 convolution (NHWC; src/models/sdxl/denoiser.py:511-551 with trainable
 * weights): dw[Cout][tap][Cin] (+)= sum over pixels of dy[pixel][Cout] * x[pixel + tap offset][Cin], the im2col implicit
 * (shifted row reads, image borders masked).  Same kernel, scratch and ordering guarantees as vft_linear_wgrad.
 * W % 8 == 0, Cin % 8 == 0, Cout % 8 == 0. */
int64_t vft_conv3x3_wgrad_scratch_floats(int64_t pixels, int Cout, int Cin);
int vft_conv3x3_wgrad(const void* dy, const void* x, void* dw, float* scratch, int B, int H, int W, int Cin, int Cout,
                      int accumulate, int out_f32, void* stream);

/* Affine-parameter gradients of the same norms (weight / bias trainable: the full fine-tune; the ATen chain autograd
 * runs for nn.LayerNorm's weight/bias in src/models/sdxl/denoiser.py:250-252):
 *   dw[c] = sum_rows dy*(1+scale)*xhat, db[c] = sum_rows dy*(1+scale); either output may be NULL.
 * scratch: vft_norm_bwd_params_scratch_floats(rows, C) floats.  out_f32: 0 = bf16 outputs, 1 = fp32.
 * Atomic-free, fixed summation order (bit-reproducible). */
int64_t vft_norm_bwd_params_scratch_floats(int64_t rows, int C);
int vft_norm_bwd_params(const void* dy, const void* x, const void* scale, const float* mean, const float* rstd,
                        void* dw, void* db, float* scratch, int64_t rows, int C, int64_t ldx,
                        int rows_per_batch, int kind, int out_f32, int accumulate /* dw, db += */, void* stream);

/* out[c] (+)= sum over rows of dy[row][c]: the bias gradient of a trainable nn.Linear / nn.Conv2d (what autograd runs as
 * aten::sum; src/models/sdxl/denoiser.py:74,167,183-204,511-551 with trainable biases).  scratch:
 * vft_norm_bwd_params_scratch_floats(rows, C) floats.  Ordered partial sums, bit-reproducible. */
int vft_colsum(const void* dy, void* out, float* scratch, int64_t rows, int C, int64_t ld, int accumulate, int out_f32,
               void* stream);
/* out[b][c] = sum over the rows of batch element b of dy[b][row][c] (dy dense [B][rows][ld]): the gradient of a per-sample channel
 * vector broadcast over the spatial positions — the ResidualBlock's time-embedding projection, src/models/sdxl/denoiser.py:524-551
 * (autograd: aten::sum over H, W).  One launch pair for the whole batch; scratch: B x vft_norm_bwd_params_scratch_floats(rows, C). */
int vft_colsum_batched(const void* dy, void* out, float* scratch, int B, int64_t rows, int C, int64_t ld, int out_f32, void* stream);

/* ---- per-head q/k norm + RoPE ---------------------------------------------------------------
 * x: (B,S,H,D) bf16 addressed by x_strides[3] = {batch, seq, head} (so views into a fused qkv
 * projection work), y/dx: (B,S,H,D) contiguous.  kind: 0 LayerNorm without affine (AuraFlow,
 * auraflow/denoiser.py:165-174), 1 RMSNorm * weight[D] (Lumina2, lumina2/denoiser.py:96-97),
 * 2 none.  cos/sin: fp32 (S, D/2) [cs_batch_stride = 0] or (B, S, D/2); NULL = no rotation.
 * Rotation = interleaved pairs (rope.py:83-111) == complex multiply (lumina2/denoiser.py:103-112).
 */
int vft_qk_norm_rope_fwd(const void* x, void* y, const void* weight, const float* cos, const float* sin,
                         float* mean, float* rstd, int B, int S, int H, int D, const int64_t* x_strides,
                         int64_t cs_batch_stride, int kind, float eps, void* stream);
int vft_qk_norm_rope_bwd(const void* dy, const void* x, const void* weight, const float* cos,
                         const float* sin, const float* mean, const float* rstd, void* dx, int B, int S,
                         int H, int D, const int64_t* x_strides, const int64_t* dy_strides /* {batch, seq, head} in
                         elements, multiples of 8; NULL = contiguous (B,S,H,D) */, int64_t cs_batch_stride, int kind,
                         void* stream);

/* ---- GroupNorm (+SiLU), NCHW ---------------------------------------------------------------
 * Replaces GroupNorm -> SiLU of the SDXL ResidualBlock (sdxl/denoiser.py:511-551) and the
 * SpatialTransformer norm (:303).  stats/bstats: fp32 [N*G*2], written by the call (stats is what backward
 * reads); dweight/dbias: optional fp32 [C], written by the call.  scratch: fp32
 * [vft_groupnorm_scratch_floats(N, C, HW, G, nhwc, affine)] for the partial sums — every reduction is two-stage with
 * a fixed order, so results are bit-reproducible (no floating-point atomics; torch's checkpoint recompute,
 * sdxl/denoiser.py:592-605, sees identical statistics).  Accesses are 16 bytes wide when H*W is a multiple of 8.
 */
int64_t vft_groupnorm_scratch_floats(int N, int C, int HW, int G, int nhwc, int affine);
int vft_groupnorm_fwd(const void* x, void* y, const void* weight, const void* bias, float* stats,
                      float* scratch, int N, int C, int HW, int G, float eps, int fuse_silu, void* stream);
int vft_groupnorm_bwd(const void* x, const void* dy, const float* stats, const void* weight,
                      const void* bias, void* dx, float* bstats, float* scratch, float* dweight,
                      float* dbias, int N, int C, int HW, int G, float eps, int fuse_silu, void* stream);
/* Channels-last variants: x / y / dy / dx are [N][HW][C] (C % 8 == 0, C <= 2560), everything else as
 * above, same statistics layout.  The SDXL UNet runs NHWC end to end on MI355X: the convolutions are
 * NHWC-native and (N, HW, C) is the SpatialTransformer's token layout (sdxl/denoiser.py:303-330), so the
 * reference's permute + reshape copies around every transformer disappear. */
int vft_groupnorm_nhwc_fwd(const void* x, void* y, const void* weight, const void* bias, float* stats,
                           float* scratch, int N, int C, int HW, int G, float eps, int fuse_silu,
                           void* stream);
int vft_groupnorm_nhwc_bwd(const void* x, const void* dy, const float* stats, const void* weight,
                           const void* bias, void* dx, float* bstats, float* scratch, float* dweight,
                           float* dbias, int N, int C, int HW, int G, float eps, int fuse_silu,
                           void* stream);

/* ---- gated activations / gated residual ------------------------------------------------------
 * GEGLU: out[M,F] = proj[:, :F] * gelu(proj[:, F:])   (sdxl/denoiser.py:183-186, exact erf GELU)
 * SwiGLU: out = silu(a) * b                            (auraflow/denoiser.py:113-119, lumina2 :212-219)
 * gate_residual: out = residual + f(gate[b,:]) * y, f = tanh (lumina2 :303,311) or identity
 *                (auraflow :408-422); residual may be NULL (then out = f(gate)*y, used in backward).
 */
int vft_geglu_fwd(const void* proj, void* out, int64_t M, int F, void* stream);
int vft_geglu_bwd(const void* proj, const void* dy, void* dproj, int64_t M, int F, void* stream);
int vft_swiglu_fwd(const void* a, const void* b, void* out, int64_t n, void* stream);
int vft_swiglu_bwd(const void* a, const void* b, const void* dy, void* da, void* db, int64_t n, void* stream);
int vft_gate_residual(const void* residual, const void* gate, const void* y, void* out, int64_t rows,
                      int C, int rows_per_batch, int use_tanh, void* stream);

/* ---- NF4-weight linear (QLoRA base) on the fused-linear kernel ------------------------------------------
 * transpose_w = 0: y[M,out] = x[M,in] . W^T (+bias) (+ ext_x[M,r] . ext_w[out,r]^T) (+residual)
 * transpose_w = 1: y[M,in]  = x[M,out] . W   (+ ext ...)   -- the data gradient
 * W = bitsandbytes NF4 payload [out*in/2] bytes (bnb.py:37-129 layout, blocksize 64), absmax = one fp32 per
 * 64 weights (a double-quantised checkpoint's absmax is expanded once at load time), code = the 16-entry map.
 * Only 4-bit codes + absmax persist in HBM.  Two forms of the dequantisation, both inside the GEMM launch:
 *   workspace != NULL (round 4, what the product passes): the launch dequantises the weight ONCE — every workgroup of the
 *     persistent grid converts its share of the codes into the workspace (bf16 image of W, or of W^T for the data
 *     gradient), the grid meets at a barrier, and the bf16 main loop of the second-generation kernel runs on the image
 *     (which stays in the 256 MB Infinity Cache: <= 28 MB for Lumina2's largest layer).  Bit-identical to vft_linear_fwd on
 *     vft_nf4_dequant's output.  The workspace (vft_linear_nf4_workspace bytes, 256-byte aligned) belongs to the caller, may
 *     be shared by every NF4 layer of a stream, and its FIRST 16 KiB (barrier words) must be zero before the first
 *     launch; launches only advance them (graph replay needs no memset).  Word 2 is an error flag: a launch that could
 *     not assemble its grid at the barrier within a bounded wait sets it instead of hanging.
 *   workspace == NULL, or a shape the second-generation kernel does not take (K % 64 != 0, r = 32, operands >= 2 GiB):
 *     dequantisation in the staging threads of the first-generation kernel's K-loop (round 1-3).
 * in_features % 64 == 0 (and out_features % 64 == 0 for transpose_w); r in {4, 8, 16, 32}.
 */
int64_t vft_linear_nf4_workspace(int out_features, int in_features);
int vft_linear_nf4(const void* x, void* y, const void* bias, const void* residual, const void* ext_x,
                   const void* ext_w, const void* packed, const float* absmax, const float* code,
                   int out_features, int in_features, int M, int r, int64_t ldx, int64_t ldy,
                   int64_t ldres, int transpose_w, void* workspace, int64_t ws_bytes, void* stream);

/* ---- 3x3 convolution (stride 1, padding 1), channels-last, implicit GEMM ------------------------------
 * y[B,H,W,Cout] = conv3x3(x[B,H,W,Cin], w[Cout,3,3,Cin]) + bias.  w is the storage of a channels-last
 * nn.Conv2d weight.  Replaces the 3x3 convolutions of the SDXL ResidualBlock / Upsample
 * (sdxl/denoiser.py:466-551); the data gradient is the same call with w' [Cin,3,3,Cout] = w flipped in
 * (kh,kw) and transposed.  Cin % 32 == 0, Cout % 8 == 0, 16-byte aligned dense tensors.
 */
int vft_conv3x3_fwd(const void* x, const void* w, void* y, const void* bias, int B, int H, int W, int Cin,
                    int Cout, void* stream);
/* The same at STRIDE 2 (padding 1): y[B,H/2,W/2,Cout] from x[B,H,W,Cin]; H, W even, Cin % 64 == 0, Cout % 8 == 0.
 * Replaces the convolution of the SDXL Downsample (sdxl/denoiser.py:371-408: nn.Conv2d(C, C, 3, stride=2, padding=1)),
 * which MIOpen ran with atomics.  Its data / weight gradients are stride-1 calls (vft_conv3x3_fwd on the flipped,
 * transposed weight / vft_conv3x3_wgrad) over the output gradient spread onto the even pixels of a zero H x W image. */
int vft_conv3x3_s2_fwd(const void* x, const void* w, void* y, const void* bias, int B, int H, int W, int Cin,
                       int Cout, void* stream);

/* ---- fused linear layer (bf16 weight) ----------------------------------------------------------
 * y[M,N] = x[M,K] . w[N,K]^T (+ bias[N]) (+ ext_x[M,r] . ext_w[N,r]^T) (+ residual[M,N])
 * One launch for the reference's LoRALinear.forward (peft/lora.py:92-104: linear(x) + lora_up(lora_down(x)) * s):
 * ext_x = lora_down(x) (vft_lora_down), ext_w = s * lora_up; and for the `x + attn(x)` / `x + ff(x)` residual
 * adds of the transformer blocks (sdxl/denoiser.py:273-284).  The data gradient dx = dy.W + (dy.B).(sA) is the
 * same call on the transposed weight.  K % 32 == 0, N % 8 == 0, strides % 8 == 0, 16-byte aligned pointers;
 * r in {4, 8, 16, 32} when the rank extension is present (ext_x / ext_w dense [.,r]).  K % 64 == 0 with r <= 16 and
 * operands below 2 GiB run on the second-generation kernel (csrc/linear8.hip), everything else on the first.
 */
int vft_linear_fwd(const void* x, const void* w, void* y, const void* bias, const void* residual,
                   const void* ext_x, const void* ext_w, int M, int N, int K, int r, int64_t ldx,
                   int64_t ldw, int64_t ldy, int64_t ldres, void* stream);
/* Tuning / A-B control of the kernel behind vft_linear_fwd and vft_conv3x3_fwd (process-wide, not thread-safe; the
 * trainers never call it): 0 = automatic choice (default), -1 = first-generation kernel (linear.hip),
 * 1 / 2 / 3 = second-generation kernel (linear8.hip) with a 256x256 / 256x128 / 128x128 output tile. */
int vft_linear_variant(int variant);
/* Per-problem tile choice measured on the target (tools/tune_lin8.py -> tuning_tables/gfx950_lin8_tiles.csv): for the
 * (M, N, K) GEMM (conv = 0) or the implicit-GEMM convolution with M = B*H*W, N = Cout, K = 9*Cin (conv = 1) use tile
 * cfg 1 / 2 / 3 instead of the analytic choice; cfg 0 removes the entry.  Process-wide; results do not depend on the
 * tile (same K order per output element), only the speed does. */
int vft_linear_tile_hint(int M, int N, int K, int conv, int cfg);

/* ---- NF4 -> bf16 materialisation ---------------------------------------------------------------
 * out[out,in] bf16 = dequantised weight in the bitsandbytes NF4 layout (src/modules/quant/bnb.py:37-129 ->
 * bnb.functional.dequantize_4bit): packed uint8 (element 2i in the high nibble), blocksize 64, absmax either fp32
 * [nblocks] (absmax_is_u8 = 0) or double-quantised uint8 [nblocks] with nested_absmax fp32 [ceil(nblocks/256)],
 * nested_code fp32 [256] and nested_offset; code = the 16-entry NF4 quant_map.  The training path never calls it
 * (vft_linear_nf4 dequantises inside the GEMM launch): compute-cache option, odd-shape fallback, tests, tooling.
 */
int vft_nf4_dequant(void* out, const void* packed, const void* absmax, int absmax_is_u8,
                    const float* nested_absmax, const float* nested_code, const float* code,
                    float nested_offset, int out_features, int in_features, void* stream);

/* dst[C][R] = src[R][C]^T, bf16 (R, C and the row strides multiples of 8).  The per-step transposed copy of a trainable
 * projection weight: autograd's data gradient dx = dy . W (torch.mm in the reference, src/models/sdxl/denoiser.py:62-64 under
 * full fine-tune) runs as vft_linear_fwd on it. */
int vft_transpose_bf16(const void* src, void* dst, int R, int C, int64_t ld_src, int64_t ld_dst, void* stream);

/* ---- mean of squares (the reduction of the MSE losses) --------------------------------------------------------
 * out[0] = sum(x[i]^2) / n over a dense fp32 buffer: with x = prediction - target this is
 * F.mse_loss(prediction, target, reduction="mean") of the reference's losses (src/modules/loss/diffusion.py:62-68,
 * src/modules/loss/flow_match.py:85-96).  Two launches (fixed chunks -> 1024 partials in `scratch` -> ordered fold), no
 * atomics, no memset node: bit-reproducible and safe to replay from a HIP graph (ATen's one-launch reduction meets through a
 * semaphore buffer zeroed by a memset node; captured in the SDXL step it returned garbage after an idle-stream synchronize).
 * scratch: vft_mean_sq_scratch_floats() floats.  x 16-byte aligned. */
int64_t vft_mean_sq_scratch_floats(void);
int vft_mean_sq(const float* x, float* out, float* scratch, int64_t n, void* stream);
/* out[0] (+)= sum(x[i]^2) over a dense bf16 buffer (accumulate != 0: added to out[0]): the squared L2 norm of a flat gradient
 * buffer = the reduction of gradient-norm clipping (ModelForTraining.after_backward -> clip_grad_norm_,
 * src/models/for_training.py:171-189).  Same scheme and scratch as vft_mean_sq. */
int vft_sum_sq_bf16(const void* x, float* out, float* scratch, int64_t n, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFT_HIP_H */
