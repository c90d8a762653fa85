/* fvhd.h - C ABI of libfvhd.so: the MI355X (gfx950) FastViTHD encode_images() path.
 *
 * The reference (apple/ml-fastvlm) is pure Python and has no FFI for this path; the boundary it
 * exposes is the duck-typed nn.Module `MobileCLIPVisionTower`
 * (llava/model/multimodal_encoder/mobileclip_encoder.py:13-116) created by `build_vision_tower`
 * (llava/model/multimodal_encoder/builder.py:6-19) and called from `encode_images`
 * (llava/model/llava_arch.py:141-144).  This header is what a binding for that boundary binds to;
 * `ml_fastvlm_amd/_lib.py` is the ctypes stub, `INTEGRATION.md` shows the reference-side patch.
 *
 * Conventions: every function returns 0 on success and a non-zero code on failure
 * (`fvhd_last_error()` gives a thread-local message); no C++ exception crosses the boundary; all
 * `void*` data arguments are DEVICE pointers unless the name says `host_`; work is enqueued on the
 * caller's HIP stream and never synchronises the device; the caller owns input/output buffers, the
 * context owns packed weights and workspace.  A context is bound to one device and is not
 * thread-safe (one context per stream/thread; the reference's callers enter `forward` one at a
 * time, llava/serve/model_worker.py:168-187).
 *
 * The multi-GPU boundary (one tower per rank + one all-gather of visual tokens) has NO entry point here on purpose: the
 * collective belongs to the host framework's process group (RCCL through torch.distributed, ml_fastvlm_amd/distributed.py);
 * the library produces the tokens a rank contributes.
 */
#ifndef FVHD_H
#define FVHD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fvhd_ctx fvhd_ctx;
typedef void* fvhd_stream_t; /* hipStream_t */

/* element types of caller-visible tensors */
#define FVHD_F32 0
#define FVHD_F16 1
#define FVHD_BF16 2

/* GEMM epilogues (fvhd_op_gemm) */
#define FVHD_EPI_NONE 0          /* out = A.W^T                                   (MHSA.qkv, mci.py:668) */
#define FVHD_EPI_BIAS 1          /* out = A.W^T + b                               (projector Linear #2)   */
#define FVHD_EPI_BIAS_GELU 2     /* out = gelu(A.W^T + b)                         (fc1, 1x1 convs, projector Linear #1) */
#define FVHD_EPI_BIAS_LS_RESID 3 /* out = resid + ls * (A.W^T + b)                (fc2 / proj + layer scale + skip) */

/* ABI version: major * 100 + minor.  A caller built against this header compares fvhd_version() with FVHD_VERSION before anything else
 * (ml_fastvlm_amd/_lib.py does): the major part changes whenever an exported signature or the meaning of an argument changes.
 * 100 = rounds 1-3; round 4 changed signatures under the same number (fvhd_op_stem_fused + w2 / b2, fvhd_op_ffn_fused / fvhd_ffn_pack +
 * precision, fvhd_op_rope + rope_theta) - a mistake this number corrects; 500 = round 5 (adds the range guard, fvhd_op_dw7_amax,
 * fvhd_op_gemm_qkv_rope / fvhd_gemm_qkv_rope_supported, the fvhd_llm_* stream contract; no signature of round 4 changed). */
#define FVHD_VERSION 500
int fvhd_version(void);
const char* fvhd_last_error(void);

/* ---- context ------------------------------------------------------------------------------------
 * Replaces MobileCLIPVisionTower.__init__/load_model (mobileclip_encoder.py:14-58): one context per
 * (device, input resolution).  `image_size` is the R of the tower name `mobileclip_l_<R>`
 * (mobileclip_encoder.py:20), a multiple of 64.  Workspace is sized for `max_batch` images and
 * grows on demand. */
int fvhd_create(fvhd_ctx** out, int device, int image_size, int max_batch);
void fvhd_destroy(fvhd_ctx* ctx);

/* Size the workspace for `max_batch` images NOW.  Growing it (here, or implicitly when a larger batch reaches fvhd_encode*)
 * synchronises the device, frees and re-allocates the arena and drops the cached graphs - the only place where the library
 * synchronises.  Implicit growth inside fvhd_encode* / fvhd_project is refused while the caller's stream is being captured;
 * fvhd_reserve itself takes no stream and must not be called while ANY stream of the device is capturing (it synchronises the
 * device, which invalidates a capture): reserve before capturing.  An arena that a caller's capture has recorded pointers into (an
 * fvhd_encode* / fvhd_project call made while the caller's stream was capturing) is retired instead of freed when a later, larger batch
 * replaces it: the caller's graph keeps replaying on valid memory until fvhd_destroy (round 5).  Every entry point
 * that takes a context runs on the context's device and restores the caller's current device before returning. */
int fvhd_reserve(fvhd_ctx* ctx, int max_batch);

/* Hand one tensor of the reference's inference-mode state dict to the library
 * (key relative to the FastViT module, e.g. "network.7.0.token_mixer.qkv.weight"; the 629 keys of
 * tests/golden/keys.json).  `host_data` is contiguous fp32 HOST memory in the reference's own
 * layout ([out,in,kh,kw] for convs, [out,in] for linears); it is copied before the call returns.
 * Replaces nn.Module.load_state_dict for the tower (model/builder.py:131). */
int fvhd_set_tensor(fvhd_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);

/* Fold eval-mode BatchNorm into the dw7x7 taps (mci.py:901-907), transpose depthwise taps to
 * [kh*kw][C], round GEMM weights to bf16, upload.  Fails if any required tensor is missing. */
int fvhd_finalize_weights(fvhd_ctx* ctx);

/* mlp2x_gelu projector weights (multimodal_projector/builder.py:23-30): host fp32,
 * w0 [hidden, mm_hidden], b0 [hidden], w2 [hidden, hidden], b2 [hidden]. */
int fvhd_set_projector(fvhd_ctx* ctx, const float* host_w0, const float* host_b0, const float* host_w2,
                       const float* host_b2, int mm_hidden, int hidden);

/* ---- hot path -----------------------------------------------------------------------------------
 * MobileCLIPVisionTower.forward_images + feature_select (mobileclip_encoder.py:60-88):
 * images [B,3,R,R] NCHW contiguous of `img_dtype` -> tokens_out [B,(R/64)^2,3072] of `out_dtype`. */
int fvhd_encode(fvhd_ctx* ctx, const void* images, int img_dtype, int batch, void* tokens_out, int out_dtype,
                fvhd_stream_t stream);

/* mm_projector forward (llava_arch.py:143): tokens [rows, mm_hidden] -> out [rows, hidden]. */
int fvhd_project(fvhd_ctx* ctx, const void* tokens, int in_dtype, int rows, void* out, int out_dtype,
                 fvhd_stream_t stream);

/* encode_images (llava_arch.py:141-144) = tower then projector, tokens kept in bf16 in workspace. */
int fvhd_encode_images(fvhd_ctx* ctx, const void* images, int img_dtype, int batch, void* out, int out_dtype,
                       fvhd_stream_t stream);

/* ---- precision of the fused ConvFFN's hidden activation ------------------------------------------
 * ConvFFN.fc1 -> GELU -> fc2 (mci.py:922-926) runs for C in {96,192,384} as ONE kernel whose hidden activation never reaches HBM.  By
 * default it is kept as gelu(x)/4 in IEEE half (FVHD_FFN_HALF: 11 mantissa bits, better than the bf16 the reference's bf16 execution
 * carries, but |fc1 output| > 262 016 SATURATES where bf16 / fp32 carry on).  The same kernel exists with an f32 GELU and a bf16 hidden
 * operand (FVHD_FFN_BF16: no range limit, ~3-7 % slower); both are compiled in and both weight images are packed, so the choice is a
 * per-block run-time switch:
 *   fvhd_set_ffn_precision / fvhd_get_ffn_precision  - by step index (fvhd_step_info); get returns -1 for a step without a fused ConvFFN.
 *     A block whose |4 * fc2.weight| would overflow f16, whose largest |fc2.weight| is below 2^-10 (f16(4 W2) would sink into the f16
 *     subnormals) or whose fc1 biases alone exceed 2^17 (range guard, below) starts as FVHD_FFN_BF16.
 *   fvhd_audit_ranges - one eager pass over `images` (a calibration batch of the deployment's real inputs) that also materialises every
 *     ConvFFN's fc1 output and reduces it to max |.|: max_abs_out[fvhd_num_steps] (0 for steps without a ConvFFN; Inf / NaN if the fc1
 *     output itself overflowed) and, when switch_above > 0, switches every fused block whose maximum exceeds it (or is not finite) to
 *     FVHD_FFN_BF16, reporting how many in *n_switched.  A margin below the 262 016 limit (the Python wrapper uses 65 504 = a factor 4)
 *     covers inputs hotter than the calibration batch.  Synchronises `stream`; not during capture.  max_abs_out / n_switched may be NULL. */
#define FVHD_FFN_HALF 0
#define FVHD_FFN_BF16 1
int fvhd_set_ffn_precision(fvhd_ctx* ctx, int step, int precision);
int fvhd_get_ffn_precision(const fvhd_ctx* ctx, int step);
int fvhd_audit_ranges(fvhd_ctx* ctx, const void* images, int img_dtype, int batch, float switch_above, float* max_abs_out,
                      int* n_switched, fvhd_stream_t stream);

/* ---- range guard (round 5): the half-precision form is never run outside its proven range twice ------
 * An audit says nothing about an image hotter than the calibration batch.  The guard is always on and costs nothing measurable: the
 * depthwise 7x7 (+BN) that produces a fused block's input A reduces max |A| on the fly (free issue slots of the matrix-core kernel),
 * and since |fc1 out_j| <= L1(W1 row j) * max|A| + |b1_j|, a block is PROVABLY inside the half-precision range while
 *     max|A| <= limit = (2^17 - max_j |b1_j|) / max_j L1(W1 row j)          (2^17 = half of the 262 016 saturation point)
 * holds (fvhd_range_guard_limit; computed from the packed weights at fvhd_finalize_weights; a block whose biases alone exceed 2^17
 * starts as FVHD_FFN_BF16).  Every fvhd_encode* call zeroes the per-step maxima, its kernels reduce into them, and the array is read
 * back asynchronously (pinned host memory + an event; no synchronisation).  The NEXT fvhd_encode* call - or fvhd_range_guard_poll -
 * compares the finished read-backs with the limits: a block over its limit is switched to FVHD_FFN_BF16 for every later call and reported.
 * The bound is sufficient, not necessary: it can move a block that would not have saturated (costing that block 3-7 %), never the other
 * way round.  What it cannot do is repair the one batch that crossed the limit - that call's output used the half form; callers that need
 * the guarantee per batch poll with wait = 1 after the call and re-encode when a step is reported (the Python tower does exactly that
 * when mm_vision_range_guard = "strict").  Inactive while the caller's stream is being captured (an event inside a graph cannot be
 * polled): graph-capturing callers calibrate with fvhd_audit_ranges first.
 * WHERE the maximum is taken (FVHD_GUARD_SITE in the environment of fvhd_create; default 0): 0 = max |A| inside the dw7x7 (+BN) kernel, limit as
 * above; 1 = one convolution earlier, max |y| of the RepMixer output inside the dw3x3 kernel, with |A_c| <= L1(folded 7x7 taps of c) max|y| +
 * |folded BN bias_c|, i.e. limit_y = (limit - max_c |bias_c|) / max_c L1(taps_c) - a looser bound (by the 7x7 taps' L1 norm, 2-7x: it moves
 * blocks to the slower form sooner than needed).  Measured on one box, whole step at B = 32 (profiles/r05_guard_cost_ab.log): guard on
 * (site 0) 23.76 / 23.77 ms, off 23.71 / 23.69 = 0.27 %; site 1 23.62 (one run).
 *   fvhd_set_range_guard(ctx, 0 / 1)   - default 1 (environment: FVHD_RANGE_GUARD=0).
 *   fvhd_range_guard_limit             - the limit on the tracked maximum (max|A| / max|y|) of a fused step (INFINITY if the weights are all
 *                                        zero, < 0 if the block can never run the half form).
 *   fvhd_range_guard_poll              - consume the finished read-backs (wait != 0: all outstanding ones, synchronising on their events);
 *                                        steps_out / amax_out [max_out] receive the steps switched since the last poll and the max|A| that
 *                                        did it, *n_out how many; with max_out <= 0 nothing is handed out and nothing forgotten: *n_out = the
 *                                        number waiting (a count query).
 * Run-ahead: four read-backs can be in flight; a fifth fvhd_encode* call whose predecessors' read-backs nobody has consumed waits on the host
 * for the oldest one's event (an asynchronous caller runs at most four encodes ahead of the GPU). */
int fvhd_set_range_guard(fvhd_ctx* ctx, int on);
int fvhd_range_guard_limit(const fvhd_ctx* ctx, int step, float* limit_out);
int fvhd_range_guard_poll(fvhd_ctx* ctx, int wait, int* steps_out, float* amax_out, int max_out, int* n_out);

/* geometry helpers (mobileclip_encoder.py:106-116) */
int fvhd_num_tokens(const fvhd_ctx* ctx);   /* (R/64)^2 */
int fvhd_hidden_size(const fvhd_ctx* ctx);  /* 3072     */

/* ---- step-level execution (parity tests / debugging) ---------------------------------------------
 * The tower is a list of "steps", one per forward() of a reference module on the running activation, in
 * the execution order of FastViT.forward (mci.py:1427-1451): step 0 = convolutional_stem, then per stage
 * [RepCPE], every RepMixerBlock / AttentionBlock, [PatchEmbed], and last conv_exp (+SE+GELU).
 * fvhd_step_info: kind 0 stem, 1 RepCPE, 2 RepMixerBlock, 3 AttentionBlock, 4 PatchEmbed, 5 conv_exp;
 * (c_in,h_in) / (c_out,h_out) = channels and side of the NHWC activation entering / leaving the step.
 * fvhd_run_steps runs steps first..last (inclusive) on x_in and writes the result to x_out, both NHWC
 * bf16 device buffers ([batch,h,h,c]; for first == 0 x_in is the NCHW bf16 image batch, for the last
 * step x_out is the [batch,T,3072] bf16 token tensor).  This is how the tests feed every block the
 * oracle's input for that block ("teacher forcing") instead of comparing only after 44 blocks. */
int fvhd_num_steps(const fvhd_ctx* ctx);
int fvhd_step_info(const fvhd_ctx* ctx, int step, int* kind, int* stage, int* block, int* c_in, int* h_in,
                   int* c_out, int* h_out);
int fvhd_run_steps(fvhd_ctx* ctx, int first, int last, const void* x_in, int batch, void* x_out,
                   fvhd_stream_t stream);

/* ---- measurement --------------------------------------------------------------------------------
 * With profiling on, every kernel launch inside fvhd_encode/fvhd_project is bracketed by HIP events
 * on the caller's stream.  fvhd_profile_read synchronises those events and returns, per kernel
 * class, the accumulated milliseconds and launch count since the last reset.  Class names:
 * "stem", "dw3", "dw7", "dw_down", "gemm_fc1", "gemm_fc2", "gemm_1x1", "gemm_qkv", "gemm_proj",
 * "layernorm", "attention", "head", "projector", "ffn_fused". */
int fvhd_profile_enable(fvhd_ctx* ctx, int on);

/* Precision option of the MHSA core (mci.py:670-679) for BASELINE.json configs[4] ("fp8 MFMA attention path"), OPT-IN:
 * on != 0 runs QK^T and PV with OCP e4m3 operands (fvhd_op_attention_fp8) in every AttentionBlock of fvhd_encode /
 * fvhd_run_steps; default 0 = bf16 operands (the parity path).  The reference has no such switch: its attention runs in
 * the tower dtype (mobileclip_encoder.py:85).  Also settable with the environment variable FVHD_ATTN_FP8=1 at fvhd_create.
 * Measured: no faster than bf16 on gfx950 (the non-scaled fp8 MFMA issues at the bf16 rate; DESIGN.md "fp8"). */
int fvhd_set_attention_fp8(fvhd_ctx* ctx, int on);

/* Kernel selection.  Default (0): every launch takes the fastest kernel for its shape INCLUDING the batch - below ~0.75 workgroups
 * per CU the depthwise 7x7 runs on the VALU kernel instead of the matrix-core one and ConvFFN as two tiled GEMMs instead of the
 * fused kernel (B = 1 at 1024^2: 4.3 -> 3.5 ms).  Results are then bit-identical for a given batch size (any order, any
 * neighbours) and equal across batch sizes only to bf16 rounding.  on != 0: the choice depends on the shape of ONE image only, so
 * an image produces the same bits in whatever batch it travels (dynamic batching with reproducible outputs).  The reference
 * makes no such promise either way (cuDNN / MIOpen pick algorithms by shape). */
int fvhd_set_batch_invariant(fvhd_ctx* ctx, int on);

/* hipGraph replay: on != 0 makes fvhd_encode / fvhd_encode_images capture the interior steps of the tower (everything between
 * the stem, which reads the caller's images, and the head, which writes the caller's buffer: ~170 launches)
 * into one hipGraph per (batch, options) on the second call with that batch size and replay it from then on - the launch-bound
 * small-batch case (TTFT, B = 1..8).  The reference's analogue is none (eager PyTorch, mobileclip_encoder.py:70-88).
 * A caller that is itself stream-capturing gets plain launches.  Also FVHD_GRAPH=1 at fvhd_create.  Default off. */
int fvhd_set_graph(fvhd_ctx* ctx, int on);
int fvhd_profile_reset(fvhd_ctx* ctx);
int fvhd_profile_read(fvhd_ctx* ctx, int max_classes, const char** names, double* ms, int64_t* launches,
                      int* n_classes);

/* ---- single ops (unit-test entry points; device pointers, packed layouts as documented) ---------- */
/* depthwise conv, NHWC bf16: x [B,H,W,Cin] -> y [B,OH,OW,Cin*mult]; w fp32 [K*K][Cout]; bias fp32 [Cout] or NULL.
 * (K,stride,mult,gelu) in {(3,1,1,0),(3,2,1,1),(7,1,1,0),(7,2,2,1),(3,1,2,0)}  - mci.py:808-811, 575-586, 921, 992-995, 442-451, 1401-1411.
 * Cin * mult must be a multiple of 32 (every FastViTHD width is: 96 ... 3072); any other combination is an error with a message. */
int fvhd_op_dwconv(fvhd_stream_t stream, const void* x, void* y, const float* w, const float* bias,
                   int B, int H, int W, int Cin, int K, int stride, int mult, int gelu);

/* The 7x7 stride-1 depthwise conv (+ bias) on the matrix cores (csrc/dwconv_mfma.hip: 16-block 4x4x4 bf16 MFMA, taps rounded to
 * bf16, fp32 accumulation) for ANY batch size - fvhd_op_dwconv / the tower take this kernel by themselves once the launch fills
 * the chip (or always, under fvhd_set_batch_invariant).  Same arguments as fvhd_op_dwconv(K = 7, stride 1, mult 1, no GELU);
 * needs C % 64 == 0 or C % 96 == 0 and W >= 16, anything else is an error. */
int fvhd_op_dw7_mfma(fvhd_stream_t stream, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int C);
/* The same convolution with the range guard's reduction (round 5): amax_bits (device, FVHD_AMAX_SLOTS = 64 32-bit words, zeroed by the caller; the
 * workgroups spread their atomics over the words, the result is the maximum of all 64) receives max |y| - over
 * everything the launch stores (VALU kernel, mfma = 0) or over the stored rows and the columns of the kernel's 64-px strips (matrix-core kernel,
 * mfma = 1: for W % 64 != 0 a superset of the image, computed from the zero padding) - as the fp32 bit pattern of a non-negative number
 * (combined with atomicMax: unsigned order = numeric order), taken from the fp32 accumulators before the rounding to bf16.  mfma = 0 is an error
 * for shapes the dispatcher gives to the matrix-core kernel, mfma = 1 for shapes that kernel does not take. */
/* RepMixer dw3x3 (+ bias) followed by the ConvFFN's dw7x7 (+ folded BatchNorm bias) in ONE launch (round 6, csrc/dwconv_fused.hip;
 * mci.py:808-811 -> :920-921): x [B,H,W,C] -> y = dw3x3(x) + b3 [B,H,W,C] (the block's residual stream, written once) and
 * a = dw7x7(y) + b7 [B,H,W,C], all NHWC bf16; w3 fp32 [9][C], w7 fp32 [49][C], b3 / b7 fp32 [C] or NULL.  Both convolutions run on the
 * 16-block 4x4x4 bf16 MFMA: the 3x3 with every tap split into two bf16 halves (16 mantissa bits - the re-parameterised centre tap is
 * 1 + eps), the 7x7 exactly as fvhd_op_dw7_mfma (taps rounded to bf16; the same bits as that entry point given the same y).
 * amax_bits: NULL or FVHD_AMAX_SLOTS words (zeroed by the caller) receiving max |a| as in fvhd_op_dw7_amax(mfma = 1).
 * Needs C % 32 == 0, C >= 64, W % 4 == 0, W >= 16; anything else is an error (C % 64 == 32 - stage 0's C = 96 - runs its last 64-channel
 * block half masked).  The tower takes this kernel for a RepMixerBlock by itself once
 * the launch fills the chip (fvhd_dw3_dw7_supported(..., 0)); never under fvhd_set_batch_invariant. */
int fvhd_op_dw3_dw7(fvhd_stream_t stream, const void* x, void* y, void* a, const float* w3, const float* b3, const float* w7,
                    const float* b7, int B, int H, int W, int C, void* amax_bits);
/* 1 when fvhd_op_dw3_dw7 takes the shape (force != 0) / when the tower picks it by itself (force == 0) */
int fvhd_dw3_dw7_supported(int B, int H, int W, int C, int force);
/* 1 when fvhd_op_dwconv runs PatchEmbed's depthwise conv (K = 7, stride 2, mult 2; mci.py:442-451) on the matrix cores (round 6,
 * csrc/dwconv_down.hip: stride 2 as two Toeplitz products over the even and the odd input pixels, taps rounded to bf16 as in fvhd_op_dw7_mfma,
 * fp32 accumulation, fp32 GELU): C_in % 32 == 0 and, for force == 0, an output map at least 24 pixels wide (narrower maps: the VALU
 * kernel's finer tiles).  A choice by shape only - the same bits whatever the batch.  The TOWER additionally keeps the VALU kernel (fp32
 * taps) for a PatchEmbed whose packed taps are not bf16 numbers (an fp32 / fp16 checkpoint): with a re-parameterised bf16 checkpoint the
 * kernel's bf16 operands are the taps themselves.  FVHD_DWDOWN_MFMA=0 in the environment keeps the VALU kernel everywhere. */
int fvhd_dw7s2_mfma_supported(int B, int H, int W, int Cin, int force);
/* the same conv on that kernel directly: x [B,H,W,Cin] -> y [B,ceil(H/2),ceil(W/2),2 Cin] NHWC bf16, w fp32 [49][2 Cin], bias fp32 [2 Cin] or
 * NULL, GELU applied; any shape with fvhd_dw7s2_mfma_supported(..., 1) */
int fvhd_op_dw7s2_mfma(fvhd_stream_t stream, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int Cin);
#define FVHD_AMAX_SLOTS 64
int fvhd_op_dw7_amax(fvhd_stream_t stream, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int C, int mfma,
                     void* amax_bits);
/* out[M,N] = epi(A[M,K] . Wt[N,K]^T): A, Wt, resid bf16; bias, ls fp32 [N]; K % 32 == 0, N % 16 == 0.  out_dtype other than bf16 only with
 * FVHD_EPI_BIAS (f16 / f32) and FVHD_EPI_NONE (f32: lm_head logits); FVHD_EPI_SWIGLU writes [M, N/2]. */
int fvhd_op_gemm(fvhd_stream_t stream, const void* A, const void* Wt, const float* bias, const float* ls,
                 const void* resid, void* out, int M, int N, int K, int epilogue, int out_dtype);
/* the residual GEMM of ConvFFN.fc2 / MHSA.proj (out = resid + ls * (A . Wt^T + bias); mci.py:926 + 1106-1109, :681 + 1185-1187) with K split over
 * `splits` workgroups per output tile - what fvhd_encode launches instead of fvhd_op_gemm(..., EPI_BIAS_LS_RESID) when fvhd_gemm_splitk_plan(M, N, K)
 * > 1 (a handful of tiles with a long K: batches of 1-8 images; never in batch-invariant mode).  partial: fp32 scratch [splits][M][N];
 * N % 128 == 0, K % (64 * splits) == 0; resid may alias out. */
int fvhd_op_gemm_splitk_ls(fvhd_stream_t stream, const void* A, const void* Wt, const float* bias, const float* ls, const void* resid, void* out,
                           float* partial, int M, int N, int K, int splits);
int fvhd_gemm_splitk_plan(int M, int N, int K);
/* LayerNormChannel (mci.py:617-623) on NHWC rows: x,y [M,C] bf16; w,b fp32 [C]. */
int fvhd_op_layernorm(fvhd_stream_t stream, const void* x, void* y, const float* w, const float* b, int M, int C, float eps);
/* MHSA core (mci.py:670-679): qkv [B*N,3C] bf16 -> out [B*N,C] bf16, head_dim 32. */
int fvhd_op_attention(fvhd_stream_t stream, const void* qkv, void* out, int B, int N, int C);
/* same, Q/K/V and P = exp(s - max) rounded to OCP e4m3 (RNE) as MFMA operands, fp32 accumulation and running maximum; the softmax
 * denominator sums the same e4m3 P values that multiply V. */
int fvhd_op_attention_fp8(fvhd_stream_t stream, const void* qkv, void* out, int B, int N, int C);
/* stem[0] (mci.py:563-574): img [B,3,R,R] of dtype -> out [B,R/2,R/2,96] bf16; w fp32 [27][96] (k = ci*9+ky*3+kx). */
int fvhd_op_stem_conv(fvhd_stream_t stream, const void* img, int dtype, void* out, const float* w, const float* bias, int B, int R);
/* convolutional_stem in one launch (mci.py:553-603): img [B,3,R,R] of dtype -> out [B,R/4,R/4,96] bf16;
 * w0 fp32 [27][96], b0 [96] as fvhd_op_stem_conv; w1 fp32 [9][96] (tap-major), b1 [96] as fvhd_op_dwconv(K=3, stride 2, gelu);
 * w2 bf16 [96][96] ([out][in], as fvhd_op_gemm takes weights), b2 fp32 [96]: stem[2], the 1x1 conv + GELU (mci.py:587-598) - or both
 * NULL: stem[0] + stem[1] only.  Bit-identical to fvhd_op_stem_conv -> fvhd_op_dwconv [-> fvhd_op_gemm(FVHD_EPI_BIAS_GELU)]. */
int fvhd_op_stem_fused(fvhd_stream_t stream, const void* img, int dtype, void* out, const float* w0, const float* b0,
                       const float* w1, const float* b1, const void* w2, const float* b2, int B, int R);
/* SEBlock + GELU of conv_exp (mci.py:72-81,198): y [B,T,C] bf16 -> out [B,T,C] of out_dtype;
 * pooled: fp32 scratch [B*(C+RD)]; scale: fp32 scratch [B,C]; wr fp32 [RD][C]; we fp32 [C][RD]; RD % 4 == 0. */
int fvhd_op_se_head(fvhd_stream_t stream, const void* y, float* pooled, float* scale, const float* wr, const float* br,
                    const float* we, const float* be, void* out, int out_dtype, int B, int T, int C, int RD);

/* Fused ConvFFN MLP (mci.py:922-926 + 1106-1109): X <- X + ls * (gelu(A.W1^T + b1).W2^T + b2), in place on X [M,C] bf16.
 * C in {96,192,384} (fvhd_ffn_fused_supported).  A [M,C] bf16; b1 fp32 [4C]; b2, ls fp32 [C];
 * w1img / w2img: DEVICE copies of the bf16 chunk images fvhd_ffn_pack writes on the host from fc1.weight [4C][C] and
 * fc2.weight [C][4C] (fp32, the reference's layouts): per chunk of 32 hidden units a 64*C-byte image in the kernel's
 * LDS byte order (XOR-swizzled 16-B slots; fc2's hidden axis permuted inside the chunk so that position 16kb+8half+j
 * holds hidden unit 16kb+8(j>>2)+4half+(j&3)).  Sizes: w1img (4C/32 + 1) * 64*C bytes (last chunk zero), w2img 4C/32 * 64*C.
 * Element types of the images by `precision`: FVHD_FFN_HALF - bf16(fc1 / 4) and IEEE half f16(4 * fc2), the kernel's hidden activation
 * is gelu(x) / 4 in f16 (11 mantissa bits instead of bf16's 8; |Phi error| <= 1.4e-3; saturates at |x| = 262016); FVHD_FFN_BF16 -
 * bf16(fc1), bf16(fc2), f32 GELU, bf16 hidden operand (no range limit).  The images are opaque to callers: pack with fvhd_ffn_pack of
 * the same library build and run them with the SAME precision. */
int fvhd_ffn_fused_supported(int C);
int fvhd_ffn_pack(int C, const float* host_fc1, const float* host_fc2, void* host_w1img, void* host_w2img, int precision);
int fvhd_op_ffn_fused(fvhd_stream_t stream, const void* A, const void* w1img, const float* b1, const void* w2img,
                      const float* b2, const float* ls, void* X, int M, int C, int precision);

/* Image preprocessing of ONE image on the device - `process_images` / `expand2square` (llava/mm_utils.py:154-184) around the
 * tower's CLIPImageProcessor (mobileclip_encoder.py:45-49): canvas of the background colour, Pillow's 8-bit bicubic resample
 * (Resample.c, bit-exact: fixed-point taps with 22 fractional bits, horizontal pass rounded to uint8, then vertical), centre crop,
 * x * (1/255).  src: uint8 HWC RGB [src_h][src_pitch bytes] sitting at (pad_top, pad_left) of the canvas, bg = 0xBBGGRR.
 * hbounds / vbounds: int32 [R][2] (first canvas column / row, tap count) and hcoef / vcoef: int32 [R][hk | vk] for the R cropped
 * output columns / rows (ml_fastvlm_amd/preprocess.py computes them like `precompute_coeffs` + `normalize_coeffs_8bpc`);
 * tmp: nrows * R * 3 bytes of scratch for the canvas rows [row0, row0 + nrows) the vertical taps touch; lut: 256 floats
 * (value * scale as the reference rounds it); out: [3][R][R] of out_dtype.  All pointers are device pointers. */
int fvhd_op_preprocess(fvhd_stream_t stream, const void* src, int src_h, int src_w, int64_t src_pitch, int pad_top, int pad_left,
                       uint32_t bg, const int32_t* hbounds, const int32_t* hcoef, int hk, const int32_t* vbounds, const int32_t* vcoef,
                       int vk, int row0, int nrows, void* tmp, const float* lut, int R, void* out, int out_dtype);

/* ---- multimodal embedding splice --------------------------------------------------
 * The data movement of LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal (llava_arch.py:233-332) as one gather:
 * out[b, t] = embedding-table row of a text token, a row of the image features, or zeros (padding), plus attention mask,
 * position ids and labels of that position.  The host side (ml_fastvlm_amd/splice.py: splice_plan) supplies, per input
 * position [B, L]: `start` = first output position of the token inside its spliced sequence (non-decreasing; dropped
 * positions carry their successor's value), `feat_row0` = first row of the image in `feats` for a -200 token, -1 for text;
 * `seqlen[b]` = length of the spliced (and truncated) sequence.  table [vocab, H], feats [n_feat_rows, H], out
 * [B, max_len, H] of `dtype`; mask_out (uint8), pos_out, labels_out (int64) [B, max_len], labels_in [B, L], each may be NULL.
 * left_pad != 0 = tokenizer_padding_side "left" (llava_arch.py:306).  All pointers are device pointers. */
int fvhd_op_splice(fvhd_stream_t stream, const int64_t* ids, const int32_t* start, const int32_t* seqlen, const int64_t* feat_row0,
                   const int64_t* labels_in, const void* table, const void* feats, void* out, uint8_t* mask_out, int64_t* pos_out,
                   int64_t* labels_out, int B, int L, int H, int max_len, int64_t vocab, int64_t n_feat_rows, int left_pad, int dtype);

/* GEMM epilogues added for the LLM prefill (fvhd_op_gemm) */
#define FVHD_EPI_RESID 4         /* out = resid + A.W^T                           (Qwen2 o_proj / down_proj + the layer's skip) */
#define FVHD_EPI_SWIGLU 5        /* out[m][j] = silu(acc[m][2j]) * acc[m][2j+1]; out is [M, N/2]; W rows interleaved gate_j, up_j */

/* ---- LLM prefill: Qwen2 decoder stack on the spliced embeddings --------------------
 * Replaces the prefill call the reference makes into the third-party `transformers` Qwen2ForCausalLM (pinned 4.48.3,
 * pyproject.toml:17): `LlavaQwen2ForCausalLM.forward` -> `super().forward(inputs_embeds=...)` (llava/model/language_model/
 * llava_qwen.py:92-103) and the first step of `generate` (:138-143).  One context per (device, model); weights arrive under the
 * model's own state-dict keys.  Arithmetic: bf16 rows, fp32 accumulation / statistics, like the tower.
 *   hidden, n_layers, n_heads, n_kv_heads, head_dim (64 | 128), intermediate, vocab, rms_eps, rope_theta = the fields of Qwen2Config
 *   (Qwen2-0.5B: 896, 24, 14, 2, 64, 4864, 151936, 1e-6, 1e6;  Qwen2-7B: 3584, 28, 28, 4, 128, 18944, 152064). */
typedef struct fvhd_llm fvhd_llm;
int fvhd_llm_create(fvhd_llm** out, int device, int hidden, int n_layers, int n_heads, int n_kv_heads, int head_dim, int intermediate,
                    int vocab, float rms_eps, float rope_theta);
void fvhd_llm_destroy(fvhd_llm* ctx);
/* One tensor of the state dict: key = "model.layers.<l>.{input_layernorm,post_attention_layernorm}.weight",
 * "model.layers.<l>.self_attn.{q,k,v}_proj.{weight,bias}", "model.layers.<l>.self_attn.o_proj.weight",
 * "model.layers.<l>.mlp.{gate,up,down}_proj.weight", "model.norm.weight", "lm_head.weight" (the embedding table when the
 * model ties them; the leading "model." may be absent).  host_data: contiguous HOST memory of `dtype` (FVHD_F32 / F16 / BF16) in the
 * reference's [out, in] layout; converted (matrices to bf16, vectors to fp32), packed (q|k|v rows concatenated, gate / up rows
 * interleaved) and uploaded before the call returns.  Any other key is an error. */
int fvhd_llm_set_tensor(fvhd_llm* ctx, const char* key, const void* host_data, int dtype, const int64_t* shape, int ndim);
/* The same tensor from DEVICE memory on the context's device (a model that already lives on the GPU): matrices must be FVHD_BF16 and
 * vectors FVHD_F32, row-major contiguous; packed by one device-to-device (2-D) copy on `stream` - no host round trip.  The caller keeps
 * dev_data alive until `stream` has run the copy.  STREAM CONTRACT (round 5): the copies are asynchronous; fvhd_llm_finalize waits (on the
 * host) for all of them, so after it returns a prefill may run on ANY stream; a tensor re-set after fvhd_llm_finalize is ordered before
 * the next fvhd_llm_prefill by an event (stream wait; a host wait when that prefill's stream is being captured). */
int fvhd_llm_set_tensor_device(fvhd_llm* ctx, const char* key, const void* dev_data, int dtype, const int64_t* shape, int ndim,
                               fvhd_stream_t stream);
int fvhd_llm_finalize(fvhd_llm* ctx);                          /* fails if a tensor is missing */
/* Qwen2Config.max_position_embeddings: rows of the rotary table (default 8192; at most 65536 rows are tabulated).  Position ids beyond
 * the table are legal: the rotary kernel then computes cos / sin itself with the table's formula - it never clamps.  Takes effect at the
 * next workspace allocation: call it before fvhd_llm_reserve / the first prefill. */
int fvhd_llm_set_max_positions(fvhd_llm* ctx, int max_position_embeddings);
int fvhd_llm_reserve(fvhd_llm* ctx, int batch, int seq_len);   /* size the workspace now (synchronises; not during stream capture) */
/* Number of times the workspace has been (re)allocated.  A prefill captured into a CALLER's hipGraph holds workspace pointers: the
 * library never frees a workspace that a capturing stream has used (a later, larger prefill allocates a new one and keeps the old one
 * alive until fvhd_llm_destroy), so such a graph stays valid; a change of this counter tells the caller that a re-capture would pick up
 * the new, larger workspace. */
int fvhd_llm_workspace_generation(const fvhd_llm* ctx);
/* Prefill: embeds [batch, seq_len, hidden] of `dtype` (the `inputs_embeds` of prepare_inputs_labels_for_multimodal / fvhd_op_splice),
 * key_valid uint8 [batch, seq_len] (its attention mask; NULL = all valid), position_ids int64 [batch, seq_len] (NULL = 0..seq_len-1)
 * -> logits_out fp32 [batch, vocab] of the LAST position of every sequence (what generate() samples the first token from).
 * k_cache / v_cache: NULL, or bf16 [n_layers][batch][n_kv_heads][seq_len][head_dim] each - the rotated keys and the values, in the
 * layout of transformers' cache layers, for a decode loop to continue from.  Enqueued on `stream`, capture-safe after fvhd_llm_reserve. */
int fvhd_llm_prefill(fvhd_llm* ctx, const void* embeds, int dtype, const uint8_t* key_valid, const int64_t* position_ids, int batch,
                     int seq_len, float* logits_out, void* k_cache, void* v_cache, fvhd_stream_t stream);
/* tests: hidden states after the last decoder layer (before the final norm) of the previous prefill, [rows, hidden] bf16 */
int fvhd_llm_debug_hidden(fvhd_llm* ctx, void* out, int rows, fvhd_stream_t stream);

/* single ops of the prefill (unit-test entry points) */
/* Qwen2RMSNorm: y = w * x * rsqrt(mean(x^2) + eps); x, y [M, H] bf16 (may alias), w fp32 [H], H % 8 == 0 */
int fvhd_op_rmsnorm(fvhd_stream_t stream, const void* x, void* y, const float* w, int M, int H, float eps);
/* apply_rotary_pos_emb (rotate_half form) in place on the q and k heads of qkv [M, (n_heads + 2 n_kv_heads) * head_dim] bf16;
 * pos int64 [M] or NULL (row % T); table fp32 [table_positions][head_dim / 2][2] = (cos, sin) - a position outside [0, table_positions)
 * is computed in the kernel from rope_theta (inv_freq_i = theta^(-2i / head_dim), fp32), never clamped; k_cache / v_cache as fvhd_llm_prefill
 * for ONE layer ([M / T][n_kv_heads][T][head_dim]) or NULL */
int fvhd_op_rope(fvhd_stream_t stream, void* qkv, const int64_t* pos, const float* table, void* k_cache, void* v_cache, int M, int T,
                 int n_heads, int n_kv_heads, int head_dim, int table_positions, float rope_theta);
/* The q|k|v projection with everything that follows it in Qwen2Attention.forward in ONE launch (round 5): out = A . Wt^T + bias rounded to bf16,
 * then - on the q and k heads of the M real rows - the rotary embedding exactly as fvhd_op_rope applies it, then the KV-cache copies; bit-identical
 * to fvhd_op_gemm(EPI_BIAS) + fvhd_op_rope.  head_dim 64 only (a wave's 64-column block of the output tile is one head, and since the round-5 tile
 * fill a lane holds both members of every rotate_half pair); fvhd_gemm_qkv_rope_supported(Mp, N, K, head_dim, n_heads, n_kv_heads) tells whether a
 * shape takes it (Mp % 128 == 0 rows incl. padding, N = (n_heads + 2 n_kv_heads) * 64, K % 64 == 0, at most one 128 x 128 tile per CU) -
 * fvhd_llm_prefill uses it when FVHD_LLM_FUSEROPE=1 is set at fvhd_llm_create (measured neutral: the launch saved comes back as epilogue time -
 * profiles/r05_ttft_fuserope_ab.log - so the default keeps the two launches).  A [Mp, K], Wt [N, K] bf16; bias fp32 [N]; the rest as fvhd_op_rope. */
int fvhd_gemm_qkv_rope_supported(int Mp, int N, int K, int head_dim, int n_heads, int n_kv_heads);
int fvhd_op_gemm_qkv_rope(fvhd_stream_t stream, const void* A, const void* Wt, const float* bias, void* out, int Mp, int N, int K, const int64_t* pos,
                          const float* table, void* k_cache, void* v_cache, int M, int T, int n_heads, int n_kv_heads, int head_dim,
                          int table_positions, float rope_theta);
/* out = resid + A . Wt^T with K split over `splits` workgroups per output tile (Qwen2 down_proj at prefill: few tiles, long K):
 * A [M, K], Wt [N, K], resid [M, N] or NULL (may alias out), out [M, N] bf16; partial: fp32 scratch [splits][M][N]; the slices are
 * summed in order (deterministic) and rounded once.  N % 128 == 0, K % (64 * splits) == 0. */
int fvhd_op_gemm_splitk(fvhd_stream_t stream, const void* A, const void* Wt, const void* resid, void* out, float* partial, int M, int N,
                        int K, int splits);
/* the same, and norm_out = Qwen2RMSNorm(out) with weight norm_w (fp32 [N]) from the reduce kernel's own pass over the finished rows - the
 * norm the next operation of a decoder layer starts with (transformers Qwen2DecoderLayer.forward: post_attention_layernorm behind o_proj,
 * input_layernorm of the next layer behind down_proj).  Both outputs are bit-identical to fvhd_op_gemm_splitk followed by fvhd_op_rmsnorm.
 * norm_out [M, N] bf16 must not alias out. */
int fvhd_op_gemm_splitk_norm(fvhd_stream_t stream, const void* A, const void* Wt, const void* resid, void* out, float* partial, int M, int N,
                             int K, int splits, const float* norm_w, void* norm_out, float eps);
/* the q|k|v projection of a decoder layer as a split-K GEMM whose reduce finishes the projection: qkv = bf16(A . Wt^T + bias) rows
 * [M, (n_heads + 2 n_kv_heads) * head_dim], then fvhd_op_rope's rotation of the q and k heads and cache copies, in one pass
 * (bit-identical to the separate steps on the same partial sums).  A [Mp, K] bf16 with Mp >= M rows readable (Mp = M rounded up to 128),
 * partial: fp32 scratch [splits][Mp][width]; width % 128 == 0, K % (64 * splits) == 0.  Replaces transformers Qwen2Attention.forward's
 * q_proj / k_proj / v_proj + apply_rotary_pos_emb + DynamicCache.update. */
int fvhd_op_qkv_splitk_rope(fvhd_stream_t stream, const void* A, const void* Wt, const float* bias, float* partial, void* qkv, const int64_t* pos,
                            const float* table, void* k_cache, void* v_cache, int M, int Mp, int K, int T, int n_heads, int n_kv_heads, int head_dim,
                            int table_positions, float rope_theta, int splits);
/* causal grouped-query attention with a key-padding mask: qkv [B*T, (n_heads + 2 n_kv_heads) * head_dim] bf16 ->
 * out [B*T, n_heads * head_dim] bf16; key_valid uint8 [B, T] or NULL; head_dim in {64, 128} */
int fvhd_op_attention_causal(fvhd_stream_t stream, const void* qkv, void* out, const uint8_t* key_valid, int B, int T, int n_heads,
                             int n_kv_heads, int head_dim);

#ifdef __cplusplus
}
#endif
#endif /* FVHD_H */
