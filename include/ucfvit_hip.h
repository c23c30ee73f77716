/*
 * ucfvit_hip.h — C ABI of libucfvit_hip.so: the MI355X (gfx950) kernels behind the UCF-VIT operator layer.
 *
 * The reference (irlyngaas/UCF-VIT) has no FFI: its drop-in boundary is the Python nn.Module operator layer
 * (src/UCF_VIT/simple/building_blocks.py: PatchEmbed:30, Mlp:94, Attention:131, Block:194; MAE gathers in
 * src/UCF_VIT/simple/arch.py:663,683).  Every entry point below replaces the torch/ATen call sites of one of
 * those operators; the replaced reference lines are cited per function.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only, no torch types.  All pointers are DEVICE pointers unless stated otherwise.
 *  - `dtype` selects the arithmetic/storage type of activations and (shadow) weights:
 *       UCFVIT_F32  : fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32)  — the reference's `simple/` mode
 *       UCFVIT_BF16 : bf16 storage, bf16 MFMA with fp32 accumulation          — the reference's fsdp MixedPrecision mode
 *    statistics (LayerNorm mean/rstd, softmax log-sum-exp), losses, parameter gradients and optimizer state
 *    are always fp32.
 *  - `stream` is a hipStream_t passed as void*; every function only enqueues work on it (no synchronisation,
 *    no allocation) so callers may capture the calls into a hipGraph.
 *  - the caller owns every buffer (inputs, outputs, workspaces).  The library keeps no tensor memory.
 *  - return value: 0 on success, <0 on error (see codes); ucfvit_last_error() returns a thread-local message.
 *    Nothing throws across the ABI and nothing calls exit().
 *  - re-entrant: may be called concurrently from the Python main thread and autograd worker threads.
 */
#ifndef UCFVIT_HIP_H
#define UCFVIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCFVIT_ABI_VERSION 12

#define UCFVIT_OK 0
#define UCFVIT_ERR_INVALID_ARGUMENT (-1)
#define UCFVIT_ERR_UNSUPPORTED (-2)
#define UCFVIT_ERR_HIP (-3)

#define UCFVIT_F32 0
#define UCFVIT_BF16 1

int ucfvit_abi_version(void);
const char* ucfvit_last_error(void);

/* Diagnostic (no reference counterpart): launches a pure bf16 MFMA stream (operands in registers, 2 waves per SIMD on every CU,
 * `iters` x 16 v_mfma_f32_16x16x32_bf16 per wave on 16 independent accumulators) and returns the FLOPs it executes (< 0: error).  Timed with HIP events by
 * bench.py: the rate is the ceiling the chip's clock / power management leaves to any bf16 MFMA kernel on that device.
 * sink: >= 256 floats of device memory (never written in practice). */
int64_t ucfvit_mfma_probe(float* sink, int iters, void* stream);

/* Diagnostic (no reference counterpart): `workgroups` workgroups that each hold one CU (512 threads, 96 KiB LDS) for about `microseconds`,
 * doing nothing — a stand-in for a collective's kernel next to the persistent GEMM grids on a one-GPU box (tools/gemm_contention.py).
 * sink: >= 512 floats of device memory (never written). */
int ucfvit_occupy(int workgroups, int microseconds, float* sink, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] = epilogue( alpha * op(A)[M,K] · op(B)[K,N] )
 *
 * Replaces nn.Linear forward/backward on the hot path: qkv / proj (building_blocks.py:150,154,159,190),
 * fc1 / fc2 (:115,119,123,127), the patch-embedding projection (conv k=s=p ≡ im2col + GEMM, :58-60,89),
 * head (arch.py:268-271,484), MAE decoder_embed / decoder_pred (arch.py:552-559,685,700).
 *
 * Operand storage ("KC" = contraction index contiguous, "KS" = contraction index strided):
 *   a_layout KC: A stored [M][K] row-major, lda = row stride      a_layout KS: A stored [K][M], lda = stride of k
 *   b_layout KC: B stored [N][K] row-major (nn.Linear weight)     b_layout KS: B stored [K][N], ldb = stride of k
 *   forward  y = x·Wᵀ      : (KC, KC)   A=x[M,K]   B=W[N,K]
 *   dgrad    dx = dy·W     : (KC, KS)   A=dy[M,N'] B=W[N',K']   (contraction over W's rows)
 *   wgrad    dW = dyᵀ·x    : (KS, KS)   A=dy[M',N] (as [K][M]), B=x[M',K'] (as [K][N])
 *
 * Epilogue, in order:  v = alpha*acc ; v += bias[n] ; act ; v += residual[m][n] ; v += C_old (accumulate) ; store.
 *   act = UCFVIT_ACT_GELU      : if aux_out, aux_out[m][n] = v (pre-activation, saved for backward); v = gelu_erf(v)
 *   act = UCFVIT_ACT_GELU_GRAD : v *= gelu_erf'(aux_in[m][n])   (dgrad through the activation; aux_in = saved pre-activation)
 *   act = UCFVIT_ACT_GELU_SAVE_DERIV / UCFVIT_ACT_MUL_AUX : the same pair with the derivative evaluated once, in the forward
 *         epilogue where gelu shares its erfc, and saved in place of the pre-activation; the backward epilogue is a multiply.
 *         The bf16 training path uses this pair (one bf16 rounding of gelu' instead of one of its argument); fp32 keeps the first.
 * `dtype` = type of A and B (and bias/residual/aux); `out_dtype` = type of C (UCFVIT_F32 for parameter gradients).
 * ------------------------------------------------------------------------------------------------------ */
#define UCFVIT_LAYOUT_KC 0
#define UCFVIT_LAYOUT_KS 1
#define UCFVIT_ACT_NONE 0
#define UCFVIT_ACT_GELU 1
#define UCFVIT_ACT_GELU_GRAD 2
#define UCFVIT_ACT_GELU_SAVE_DERIV 3 /* forward:  aux_out[m][n] = gelu_erf'(v) (required); v = gelu_erf(v) */
#define UCFVIT_ACT_MUL_AUX 4         /* backward: v *= aux_in[m][n]   (aux_in = the derivative saved by ACT_GELU_SAVE_DERIV) */

typedef struct ucfvit_gemm_desc {
    const void* A;
    const void* B;
    void* C;
    const void* bias;     /* [N] dtype, or NULL */
    const void* residual; /* [M][ldr] dtype, or NULL */
    const void* aux_in;   /* [M][ldaux] dtype, or NULL (ACT_GELU_GRAD) */
    void* aux_out;        /* [M][ldaux] dtype, or NULL (ACT_GELU) */
    int64_t M, N, K;
    int64_t lda, ldb, ldc, ldr, ldaux;
    int32_t a_layout, b_layout;
    int32_t dtype, out_dtype;
    int32_t act;
    int32_t accumulate; /* 1: C += result (gradient accumulation) */
    float alpha;
    void* workspace;         /* optional fp32 scratch for split-K partial sums (see ucfvit_gemm_workspace), or NULL */
    int64_t workspace_bytes;
    float* c_colsum_partial; /* optional by-product of the epilogue: per block of output rows, the column sums of the values written to
                              * C (taken in fp32 before the rounding to dtype): fp32 [ucfvit_gemm_colsum_rows(desc)][N], every entry
                              * written.  Summed over its rows (ucfvit_reduce_rows) it is the bias gradient of the Linear layer whose
                              * output gradient this GEMM produces (fc1: C = dh), so dh is not read again by ucfvit_colsum.  NULL: off. */
    void* sched_state;       /* optional: UCFVIT_GEMM_SCHED_BYTES of device memory, 16-byte aligned, zeroed ONCE by the caller and then
                              * used by every launch of ONE stream.  With it the persistent 256x256 kernel hands its output tiles out through
                              * device-scope atomic counters (work-conserving when another kernel — an RCCL collective overlapping backward —
                              * holds some of the CUs: workgroups that start late find the list empty instead of owning a share of it) and
                              * leaves the state zeroed.  Results do not depend on the tile order (every tile is computed by one workgroup either way); a launch
                              * with sched_state always takes the ping-pong kernel, a launch without it may take the staggered kernel (csrc/gemm_stagger.hip),
                              * whose rows 128-255 of a tile accumulate over K in a rotated — equally fixed — order: the same values up to fp32 summation order.
                              * NULL: static tile order. */
} ucfvit_gemm_desc;
#define UCFVIT_GEMM_SCHED_BYTES 1024

/* bytes of workspace the split-K path would use for this problem (0: none).  Without it the GEMM still runs, un-split. */
int64_t ucfvit_gemm_workspace(const ucfvit_gemm_desc* desc);
int ucfvit_gemm(const ucfvit_gemm_desc* desc, void* stream);
/* rows of desc->c_colsum_partial this problem would write (desc->c_colsum_partial itself is not read); 0: the kernel that runs this
 * problem has no such by-product and ucfvit_gemm rejects a non-NULL c_colsum_partial with UCFVIT_ERR_UNSUPPORTED. */
int64_t ucfvit_gemm_colsum_rows(const ucfvit_gemm_desc* desc);
/* out[n] (+)= sum_r partial[r][n]  (fp32, fixed order) — the second stage of every two-stage column sum of this library */
int ucfvit_reduce_rows(const float* partial, float* out, int64_t rows, int64_t N, int accumulate, void* stream);

/* n <= 32 epilogue-free GEMMs with the same K, layouts and dtypes (the weight gradients dW_qkv, dW_proj, dW_fc1, dW_fc2 of one
 * or several transformer Blocks all contract over the B*N tokens) as ONE persistent launch over the union of their output tiles:
 * fills the 256 CUs without split-K partial sums (ViT-L: 192 tiles per Block, so four Blocks = 768 tiles = 3 whole rounds).
 * Falls back to n ucfvit_gemm calls when the set is not groupable. */
int ucfvit_gemm_grouped(const ucfvit_gemm_desc* descs, int64_t n, void* stream);

/* column sums  out[n] (fp32) (+)= sum_m x[m][n]   — bias gradients of every nn.Linear (autograd of :159,190,123,127) */
int64_t ucfvit_colsum_workspace(int64_t M, int64_t N); /* bytes of fp32 scratch for the deterministic two-stage sum */
int ucfvit_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int accumulate, void* workspace, int dtype,
                  void* stream);

/* ------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (biased variance, affine): nn.LayerNorm(D, eps) at building_blocks.py:212,226,
 * arch.py:170,266 (eps 1e-6) and arch.py:560 (decoder_norm, eps 1e-5).
 * x,y,dy,dx: [rows][D] dtype ; gamma,beta: [D] dtype ; mean,rstd: [rows] fp32 (saved for backward).
 * bwd: dx = LN-grad(dy) + dres (dres: optional [rows][D] dtype, the residual branch's gradient, fused; NULL = none);
 *      dgamma/dbeta are fp32 [D]; `accumulate` adds into them.  workspace: fp32, >= ucfvit_layernorm_bwd_workspace() bytes.
 *      dx_colsum: optional fp32 [D] (+)= column sums of dx (taken in fp32 before rounding): dx is the gradient of the residual
 *      stream, i.e. the output gradient of the proj / fc2 Linear before this norm, so this IS that layer's bias gradient and the
 *      separate ucfvit_colsum pass over dx is not needed.  NULL = off.
 * ------------------------------------------------------------------------------------------------------ */
int ucfvit_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                         int64_t rows, int64_t D, float eps, int dtype, void* stream);
int64_t ucfvit_layernorm_bwd_workspace(int64_t rows, int64_t D);
int ucfvit_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                         const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t D, int accumulate,
                         float* dx_colsum, int dx_colsum_accumulate, void* workspace, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused multi-head self-attention core, softmax(q·kᵀ·scale)·v, non-causal, no mask, dropout 0:
 * building_blocks.py:159-189 (reshape [B,N,3,H,dh] → permute(2,0,3,1,4) → SDPA / explicit math → transpose → reshape).
 * qkv : [B][N][3][H][dh] dtype — exactly the output of the qkv Linear (no permute copy is made)
 * out : [B][N][H*dh] dtype     — already in the layout `proj` consumes
 * lse : [B][H][N] fp32, log2-domain log-sum-exp of the scaled scores (saved for backward)
 * bwd : dqkv has qkv's layout; delta_ws is fp32 [B][H][N] scratch.  dh ∈ {32, 64, 128}.
 * ------------------------------------------------------------------------------------------------------ */
int ucfvit_attention_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, int64_t dh,
                         float scale, int dtype, void* stream);
int ucfvit_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                         float* delta_ws, int64_t B, int64_t N, int64_t H, int64_t dh, float scale, int dtype,
                         void* stream);
/* The same backward, also handing out what the qkv bias gradient needs (building_blocks.py:154 qkv = nn.Linear(dim, 3 dim, bias=qkv_bias): its
 * bias gradient is dqkv summed over all B N token rows) so that dqkv is not read a second time for it:
 *   colsum_partial fp32 [B][2][H][dh]: row b holds the column sums of dQ over batch element b's N tokens (from the fp32 gradients before they
 *   are rounded), then H dh zeros for dK; the caller sums the B rows (ucfvit_reduce_rows) into the first two thirds of the bias gradient.
 *   The K third of the bias gradient is identically zero — sum_key dS[q][key] = sum_key P (dP - delta) = delta - delta: a key bias shifts
 *   every score of a row alike and the softmax does not see it (the reference's value there is rounding noise around 0) — and the V third
 *   is the column sum of `dout` (sum_key P = 1), which the projection's data-gradient GEMM can produce in its epilogue
 *   (ucfvit_gemm c_colsum_partial): neither needs this kernel.
 * Only where the fused short-sequence kernel runs: ucfvit_attention_bwd_colsum_supported returns 1 (bf16, head dim 32 / 64, N <= 256),
 * else 0 and ucfvit_attention_bwd_colsum fails with UCFVIT_ERR_UNSUPPORTED. */
int ucfvit_attention_bwd_colsum_supported(int64_t B, int64_t N, int64_t H, int64_t dh, int dtype);
int ucfvit_attention_bwd_colsum(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                float* delta_ws, float* colsum_partial, int64_t B, int64_t N, int64_t H, int64_t dh, float scale,
                                int dtype, void* stream);

/* Attention of a QUERY block against ANOTHER token block's keys / values: the building block of ring sequence parallelism (no reference
 * counterpart: the reference constructs seq_par_group and asserts seq_par_size == 1, training_scripts/train_masked_fsdp.py:220).
 *   q [B][Nq][ldq], k / v [B][Nk][ldkv] with head h at columns [h*dh, (h+1)*dh);  out [B][Nq][H*dh] (dtype), lse [B][H][Nq] in log2 units.
 * Backward of one (query block, key block) pair of a softmax that spans several key blocks: `lse` is the log-sum-exp over ALL key blocks and
 * `out` the final merged output (delta = rowsum(dO * O) is taken from them), so dq / dk / dv (fp32 [B][Nq|Nk][H*dh], += when accumulate)
 * receive exactly this pair's terms.  delta_ws: B*H*Nq floats of scratch. */
int ucfvit_attention_cross_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t Nq, int64_t Nk, int64_t H,
                               int64_t dh, int64_t ldq, int64_t ldkv, float scale, int dtype, void* stream);
int ucfvit_attention_cross_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, float* dq,
                               float* dk, float* dv, float* delta_ws, int64_t B, int64_t Nq, int64_t Nk, int64_t H, int64_t dh, int64_t ldq,
                               int64_t ldkv, float scale, int accumulate, int dtype, void* stream);
/* online merge of partial results over disjoint key blocks: lse' = log2(2^lse_acc + 2^lse_part), o' = o_acc 2^(lse_acc - lse') +
 * o_part 2^(lse_part - lse'); o_acc fp32 [B][Nq][H*dh], o_part dtype; first != 0: o_acc = o_part, lse_acc = lse_part */
int ucfvit_attention_merge(float* o_acc, float* lse_acc, const void* o_part, const float* lse_part, int64_t B, int64_t Nq, int64_t H, int64_t dh,
                           int first, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Patch embedding front end (PatchEmbed.forward, building_blocks.py:78-92): non-overlapping p×p(×p) patches of an
 * NCHW / NCHWD fp32 image are laid out as GEMM rows [B·L][C·p^nd] with K-order (c, ph, pw[, pd]) = the flattening of
 * the conv weight [D,C,p,p(,p)], token order row-major over the patch grid.  Coalesced reads of image rows through
 * LDS tiles; output in `dtype`.  nd = 2 or 3; dims = {H, W} or {H, W, Dz}.
 * ------------------------------------------------------------------------------------------------------ */
int ucfvit_im2col(const float* img, void* cols, int64_t B, int64_t C, const int64_t* dims, int nd, int64_t p,
                  int dtype, void* stream);

/* Token assembly (VIT._pos_embed, arch.py:367-393): out[b][0] = cls + pos[0] (if cls), out[b][t+pre] = patches[b][t] + pos[t+pre].
 * pos may be NULL (pos_embed='none').  bwd: dpatches = dout[:,pre:], dpos (fp32,[N][D]) (+)= sum_b dout, dcls (fp32,[D]) (+)= sum_b dout[:,0]. */
int ucfvit_tokens_fwd(const void* patches, const void* cls, const void* pos, void* out, int64_t B, int64_t L, int64_t D,
                      int has_cls, int dtype, void* stream);
int ucfvit_tokens_bwd(const void* dout, void* dpatches, float* dpos, float* dcls, int64_t B, int64_t L, int64_t D,
                      int has_cls, int accumulate, int dtype, void* stream);

/* Adaptive-patching front end (VIT.forward_features / _pos_embed with adaptive_patching=True, arch.py:465-467, :366-393).
 * The data loader delivers the token sequence already cut and resized: x fp32 [B][C][S][P] (S tokens of P = p^nd pixels per
 * channel) and seq_ps fp32 [B][S][kin] (kin = 3 for 2-D, 4 for 3-D input: position and size of each token).
 * ucfvit_seq_patches: rows_out[b*S+s][p*C+c] = x[b][c][s][p]  (einops 'b c s p -> b s (p c)'), in `dtype`; C*P <= 16384.
 * ucfvit_adaptive_pos_fwd: out[b][0] = cls (if has_cls; its position embedding is zero, arch.py:381-385),
 *     out[b][t+pre] = x[b][t] + GELU(seq_ps[b][t] . w^T + bias)     (adaptive_pos_dep_emb = Linear(kin, D) + erf-GELU, arch.py:311-321)
 *     x [B*S][D], w [D][kin], bias [D], cls [D], out [B][S+pre][D], all `dtype`; the pre-activation is recomputed in backward.
 * ucfvit_adaptive_pos_bwd: dx [B*S][D] = dout[:,pre:] (or NULL), dw fp32 [D][kin], dbias fp32 [D], dcls fp32 [D] (each may be NULL);
 *     accumulate is a bit mask (1: dw +=, 2: dbias +=, 4: dcls +=).  workspace: ucfvit_adaptive_pos_bwd_workspace bytes.
 *     Deterministic: per-chunk partial sums in the workspace, then one ordered pass. */
int ucfvit_seq_patches(const float* x, void* rows_out, int64_t B, int64_t C, int64_t S, int64_t P, int dtype, void* stream);
int ucfvit_adaptive_pos_fwd(const void* x, const float* seq_ps, const void* w, const void* bias, const void* cls, void* out,
                            int64_t B, int64_t S, int64_t D, int kin, int has_cls, int dtype, void* stream);
int64_t ucfvit_adaptive_pos_bwd_workspace(int64_t B, int64_t S, int64_t D, int kin, int has_cls, int dtype);
int ucfvit_adaptive_pos_bwd(const void* dout, const float* seq_ps, const void* w, const void* bias, void* dx, float* dw, float* dbias,
                            float* dcls, int64_t B, int64_t S, int64_t D, int kin, int has_cls, int accumulate, void* workspace,
                            int dtype, void* stream);

/* GPU-side fixed-length quadtree patcher (the adaptive-patching data transform: dataloaders/quadtree.py:84-174 FixedQuadTree,
 * dataloaders/transform.py:9-55 Patchify), a whole batch per call.  Edge detection is not included: `edges` is its result.
 * ucfvit_quadtree_build: edges uint8 [B][H][W] (cv2.Canny output: 0 / 255).  Per image the reference's greedy refinement: replace
 *     the FIRST node of maximum value (value = sum(region) / 255) by its quadrants lt, rt, lb, rb in place until fixed_length = L
 *     nodes exist (L must be 3n+1, train_unetr_simple.py:214) or that node is 2 pixels wide.  Bit-exact integer logic.
 *     nodes int32 [B][L][4] = (x1, x2, y1, y2) in list order, values int32 [B][L], count int32 [B] (valid nodes; the rest is padding),
 *     seq_ps fp32 [B][L][3] = (size = x2-x1, centre x, centre y); padding: size 0, centre (-1, -1) like FixedQuadTree.serialize.
 *     workspace: ucfvit_quadtree_workspace bytes (summed-area tables).  H, W < 32768, H*W*255 < 2^32.
 * ucfvit_quadtree_serialize: img fp32 [B][H][W][C] (channels last, as the reference's numpy images); every node's region resampled
 *     to p x p with the cv2.INTER_CUBIC kernel (A = -0.75, aligned pixel centres, replicated border) into seq fp32 [B][L][p][p][C];
 *     Patchify's plain reshape of that memory to [C][L][p*p] per image is the model input x[B][C][S][P].  Padding nodes: zeros. */
int64_t ucfvit_quadtree_workspace(int64_t B, int64_t H, int64_t W);
int ucfvit_quadtree_build(const uint8_t* edges, int32_t* nodes, int32_t* values, int32_t* count, float* seq_ps, int64_t B, int64_t H,
                          int64_t W, int64_t L, void* workspace, void* stream);
int ucfvit_quadtree_serialize(const float* img, const int32_t* nodes, const int32_t* count, float* seq, int64_t B, int64_t H, int64_t W,
                              int64_t C, int64_t L, int64_t p, void* stream);

/* Octree patcher for 3-D volumes (dataloaders/octree.py:66-151 FixedOctTree, transform.py:57-132 Patchify_3D's tree + serialize part).
 * domain uint8 [B][N][N][N] indexed [z][y][x] (cubic, N <= 256); value = sum(region) / norm_factor (norm_factor = int(255 / channels),
 * transform.py:120); children in the reference's order (x fastest, then y, then z); fixed_length L must be 7n+1; stop when the first
 * maximum node is 2 wide.  nodes int32 [B][L][6] = (x1, x2, y1, y2, z1, z2), values, count as for the quadtree; seq_ps fp32 [B][L][4] =
 * (size, centre x, y, z), padding size 0 / centre -1.  workspace: ucfvit_octree_workspace bytes.
 * ucfvit_octree_serialize: img fp32 [B][N][N][N][C]; leaf img[z1:z2, y1:y2, x1:x2, :] -> p^3 by linear interpolation with aligned corners
 * (the reference's scipy RegularGridInterpolator on linspace(0, n, n) -> linspace(0, n, p)), seq fp32 [B][L][p][p][p][C]. */
int64_t ucfvit_octree_workspace(int64_t B, int64_t N);
int ucfvit_octree_build(const uint8_t* domain, int32_t* nodes, int32_t* values, int32_t* count, float* seq_ps, int64_t B, int64_t N, int64_t L,
                        int norm_factor, void* workspace, void* stream);
int ucfvit_octree_serialize(const float* img, const int32_t* nodes, const int32_t* count, float* seq, int64_t B, int64_t N, int64_t C,
                            int64_t L, int64_t p, void* stream);

/* Variable aggregation (VariableMapping_Attention, simple/building_blocks.py:301-373, called by VIT.aggregate_variables,
 * simple/arch.py:414-432) with one aggregated variable: for every token row r and head h,
 *     p_v = softmax_v(scale * q_h . k[v][r][h]),  out[r][h] = sum_v p_v * v[v][r][h].
 * kv [V][R][2D] `dtype` (the kv Linear applied to the V per-variable token embeddings: k = columns [0, D), v = [D, 2D)), q fp32 [D]
 * (the q Linear applied to the learnt query), out [R][D] `dtype`, lse fp32 [R][H] (saved for backward).
 * bwd: dkv [V][R][2D], dq_rows fp32 [R][D] (its column sums = the gradient of q; ucfvit_colsum).  head_dim | D, both multiples of
 * 8 (bf16) / 4 (fp32), D <= 2048 (bf16) / 1024 (fp32), head_dim / 8 (4) a power of two. */
int ucfvit_varagg_fwd(const void* kv, const float* q, void* out, float* lse, int64_t R, int64_t V, int64_t D, int64_t head_dim, float scale,
                      int dtype, void* stream);
int ucfvit_varagg_bwd(const void* kv, const float* q, const void* out, const float* lse, const void* dout, void* dkv, float* dq_rows, int64_t R,
                      int64_t V, int64_t D, int64_t head_dim, float scale, int dtype, void* stream);

/* Softmax cross-entropy, mean over the batch (nn.CrossEntropyLoss, training_scripts/train_class_simple.py:24-30).
 * logits [B][C] dtype, labels int64 [B]; loss: fp32 scalar (device); row_loss: fp32 [B] per-sample losses (also scratch);
 * dlogits [B][C] dtype = grad_scale*(softmax - onehot)/B, or NULL. */
int ucfvit_cross_entropy(const void* logits, const int64_t* labels, float* loss, float* row_loss, void* dlogits, int64_t B,
                         int64_t C, float grad_scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * MAE random masking index math (MAE.random_masking, arch.py:663-681), bit-exact for distinct noise values:
 * ids_restore[b][i] = rank of noise[b][i] in its row (= argsort(argsort(noise))), ids_shuffle[b][r] = argsort(noise)[r]
 * (ids_keep = ids_shuffle[:, :len_keep]), mask[b][i] = (ids_restore[b][i] >= len_keep) ? 1 : 0.
 * int64 indices [B][L], fp32 mask [B][L], like torch.
 * ------------------------------------------------------------------------------------------------------ */
int ucfvit_mae_mask(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, float* mask, int64_t B, int64_t L,
                    int64_t len_keep, void* stream);

/* Row gather  out[b][r][:] = src[b][idx[b*idx_stride + r]][:]  (torch.gather(seq, 1, ids_keep[...,None].repeat), arch.py:675) —
 * bit-exact copy; src [B][L][D], out [B][R][D].  scatter is its adjoint for a duplicate-free idx:
 * dsrc = 0; dsrc[b][idx[b][r]][:] = dout[b][r][:]. */
int ucfvit_gather_rows(const void* src, const int64_t* idx, void* out, int64_t B, int64_t L, int64_t R, int64_t D,
                       int64_t idx_stride, int dtype, void* stream);
int ucfvit_scatter_rows(const void* dout, const int64_t* idx, void* dsrc, int64_t B, int64_t L, int64_t R, int64_t D,
                        int64_t idx_stride, int dtype, void* stream);

/* MAE un-shuffle (MAE.mask_head, arch.py:687-697): x_ = cat(x[B,R,D], mask_token repeated) ; out[b][i] = x_[b][ids_restore[b][i]] (+ pos[i]).
 * bwd: dx[b][r] = dout[b][i : ids_restore==r] ; dmask_token (fp32 [D]) (+)= sum over masked positions ; dpos (fp32 [L][D]) (+)= sum_b dout. */
int ucfvit_unshuffle_fwd(const void* x, const void* mask_token, const int64_t* ids_restore, const void* pos, void* out,
                         int64_t B, int64_t L, int64_t R, int64_t D, int dtype, void* stream);
int64_t ucfvit_unshuffle_bwd_workspace(int64_t B, int64_t D);
int ucfvit_unshuffle_bwd(const void* dout, const int64_t* ids_restore, void* dx, float* dmask_token, float* dpos,
                         int64_t B, int64_t L, int64_t R, int64_t D, int accumulate, void* workspace, int dtype, void* stream);

/* MAE reconstruction loss against patchify(img) without materialising the target (utils/misc.py:14-33 'nchpwq->nhwpqc',
 * training_scripts/train_masked_simple.py:43-47; masked variant utils/metrics.py:11-17).
 * pred [B][L][p^nd*C] dtype with per-patch order (ph,pw[,pd],c); img NCHW(D) fp32; mask fp32 [B][L] or NULL (plain MSE over all).
 * loss: fp32 scalar (device, overwritten); dpred = grad_scale * dloss/dpred (or NULL). workspace: >= 2049 floats.
 * nd = 1: the adaptive-patching target (train_masked_simple.py:24-32): img is the token sequence x fp32 [B][C][S][P], dims = {S},
 * p = P pixels per token and channel, pred [B][S][P*C] with per-token order (p, c) = rearrange 'b c s p -> b s (p c)'. */
int ucfvit_patch_mse(const void* pred, const float* img, const float* mask, float* loss, void* dpred, int64_t B, int64_t C,
                     const int64_t* dims, int nd, int64_t p, float grad_scale, float* workspace, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter segment (torch.optim.AdamW semantics; utils/misc.py:58-84):
 *   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g² ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 * grads are fp32 (grad_dtype F32) or bf16 (BF16), multiplied by grad_scale first.  If shadow != NULL the updated
 * parameter is also written as bf16 (the compute copy used by the bf16 kernels) in the same pass.
 * ------------------------------------------------------------------------------------------------------ */
int ucfvit_adamw(float* p, const void* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
                 int grad_dtype, void* stream);

/* dtype conversion / scaling helpers (fp32 master → bf16 shadow cast; bf16 gradient transport for the DP all-reduce) */
int ucfvit_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, float scale, void* stream);

/* Batched 2-D transposes of bf16 matrices inside one flat buffer: the transposed shadow of every nn.Linear weight, so the
 * data-gradient GEMMs (dx = dy·W, building_blocks.py:123,127,159,190 under autograd) read both operands contraction-contiguous.
 * table (device): int64 [n_mats][5] = {src_off, dst_off, rows, cols, first_tile} in elements; rows, cols multiples of 8;
 * a matrix uses ceil(rows/64)*ceil(cols/64) tiles; total_tiles = sum. */
int ucfvit_transpose_batched(const void* src, void* dst, const int64_t* table, int64_t n_mats, int64_t total_tiles, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * UNETR convolutional decoder, HBM-bound part.  Tensors in torch's N C (D) H W layout: `rows` = N*C contiguous rows
 * of S voxels.  Reference call sites: src/UCF_VIT/simple/arch.py:808-940 (monai UnetResBlock: conv -> instance norm -> LeakyReLU(0.01) ->
 * conv -> instance norm, + residual, LeakyReLU), training_scripts/train_unetr_simple.py:38 (monai DiceCELoss(to_onehot_y, softmax,
 * squared_pred)); monai is not vendored: parity is against the plain-torch restatement of these formulas.
 *
 * ucfvit_instnorm_fwd:  mean / rstd per row (biased variance, eps), y = lrelu((x - mean) rstd [+ res], slope)   (slope 1: no activation)
 * ucfvit_instnorm_bwd:  dn = dy (y > 0 ? 1 : slope); dres = dn (if dres); dx = rstd (dn - mean(dn) - n mean(dn n)), n = (x - mean) rstd
 * workspace: ucfvit_instnorm_workspace(rows, S) bytes (both directions).  S must be a multiple of 16 bytes / element size.
 * ------------------------------------------------------------------------------------------------------ */
int64_t ucfvit_instnorm_workspace(int64_t rows, int64_t S);
int ucfvit_instnorm_fwd(const void* x, const void* res, void* y, float* mean, float* rstd, int64_t rows, int64_t S, float eps, float slope,
                        void* workspace, int dtype, void* stream);
int ucfvit_instnorm_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, void* dx, void* dres, int64_t rows,
                        int64_t S, float slope, void* workspace, int dtype, void* stream);
/* Dice + cross-entropy of logits [B][n][S] (2 <= n <= 8) against int64 labels [B][S], fused forward + backward:
 *   p = softmax over the classes; dice_bc = 1 - (2 sum_v p onehot + smooth_nr) / (sum_v p^2 + sum_v onehot + smooth_dr);
 *   loss = mean_bc dice_bc + mean_bv -log p[label];  dlogits (dtype, may be NULL) = grad_scale * d loss / d logits. */
int64_t ucfvit_dice_ce_workspace(int64_t B, int64_t S);
int ucfvit_dice_ce(const void* logits, const int64_t* labels, float* loss, void* dlogits, int64_t B, int64_t n, int64_t S, float smooth_nr,
                   float smooth_dr, float grad_scale, void* workspace, int dtype, void* stream);

/* The same loss on strided logits: element (b, class c, voxel i) at logits[b stride_b + c stride_c + i stride_s] (dlogits alike).
 * N C (D) H W is (n S, S, 1); a channels-last tensor whose voxel rows are ld apart is (S ld, 1, ld). */
int ucfvit_dice_ce_strided(const void* logits, const int64_t* labels, float* loss, void* dlogits, int64_t B, int64_t n, int64_t S,
                           int64_t stride_b, int64_t stride_c, int64_t stride_s, float smooth_nr, float smooth_dr, float grad_scale,
                           void* workspace, int dtype, void* stream);

/* Dice + CE over a volume SHARDED across the ranks of a sequence-parallel group (X-slabs of the UNETR decoder; no reference counterpart: the
 * reference asserts seq_par_size == 1, training_scripts/train_masked_fsdp.py:220 — the loss itself is train_unetr_simple.py:38).  Every term
 * is a function of per-(batch, class) sums over voxels: ucfvit_dice_ce_stats writes this rank's sums (stats: B x ucfvit_dice_ce_stats_floats()
 * floats; workspace: ucfvit_dice_ce_workspace bytes), the caller adds them over the group, ucfvit_dice_ce_from_stats evaluates the loss of the
 * WHOLE volume from the summed stats (rewritten in place) and the gradient of the LOCAL logits (S local voxels, S_total voxels of the whole
 * volume, both per batch element).  Unsharded: S_total = S and the pair equals ucfvit_dice_ce_strided. */
int ucfvit_dice_ce_stats_floats(void);
int ucfvit_dice_ce_stats(const void* logits, const int64_t* labels, float* stats, int64_t B, int64_t n, int64_t S, int64_t stride_b, int64_t stride_c,
                         int64_t stride_s, void* workspace, int dtype, void* stream);
int ucfvit_dice_ce_from_stats(const void* logits, const int64_t* labels, float* stats, float* loss, void* dlogits, int64_t B, int64_t n, int64_t S,
                              int64_t S_total, int64_t stride_b, int64_t stride_c, int64_t stride_s, float smooth_nr, float smooth_dr,
                              float grad_scale, int dtype, void* stream);

/* Channels-last instance norm (+ LeakyReLU, + residual) for the layout of the convolution kernels below: x, res, y [B][S][C] bf16,
 * mean / rstd [B][C] fp32; C a power of two in 8..2048.  Same formulas as ucfvit_instnorm_fwd / _bwd.  had_res: the forward pass added a
 * residual, so the activation mask is read from y; without one sign(y) = sign(x - mean) and y is not read at all (dres requires had_res).
 * ld_dy: voxel-row stride of dy in elements (C for a dense tensor; larger when dy is a channel slice of a concatenation's gradient). */
int64_t ucfvit_instnorm_cl_workspace(int64_t B, int64_t S, int64_t C);
int ucfvit_instnorm_cl_fwd(const void* x, const void* res, void* y, float* mean, float* rstd, int64_t B, int64_t S, int64_t C, float eps,
                           float slope, void* workspace, void* stream);
int ucfvit_instnorm_cl_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, void* dx, void* dres, int64_t B,
                           int64_t S, int64_t C, int64_t ld_dy, float slope, int had_res, void* workspace, void* stream);

/* ucfvit_instnorm_cl_bwd in two calls, for a volume sharded across ranks: _bwd_sums leaves the per-(batch, channel) MEANS over the local voxels
 * of dy' (the gradient behind the activation mask) and of dy' xhat in m1 / m2 [B][C]; the caller averages them over the ranks (equal slabs:
 * the mean over the whole volume) and passes them to _bwd_apply.  (The forward statistics of a sharded volume combine the same way from
 * ucfvit_instnorm_cl_stats: mean = avg(mean_r), var = avg(var_r + mean_r^2) - mean^2.) */
int ucfvit_instnorm_cl_bwd_sums(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, float* m1, float* m2, int64_t B,
                                int64_t S, int64_t C, int64_t ld_dy, float slope, int had_res, void* workspace, void* stream);
int ucfvit_instnorm_cl_bwd_apply(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* m1, const float* m2,
                                 void* dx, void* dres, int64_t B, int64_t S, int64_t C, int64_t ld_dy, float slope, int had_res, void* stream);

/* The tail of a residual block whose residual branch is itself normalised (monai UnetResBlock with the 1x1x1 projection):
 *   ucfvit_instnorm_cl_stats:  mean / rstd of x only (ucfvit_instnorm_cl_fwd = this + the apply pass)
 *   ucfvit_instnorm_cl_apply2: y = lrelu((x - mean) rstd + (x2 - mean2) rstd2, slope) — the normalised branch is never materialised
 *   ucfvit_instnorm_cl_bwd2:   dx, dx2 from dy, y (activation mask) and the raw inputs: one pair of passes for both normalisations
 *                              (workspace: ucfvit_instnorm_cl_bwd2_workspace bytes). */
int ucfvit_instnorm_cl_stats(const void* x, float* mean, float* rstd, int64_t B, int64_t S, int64_t C, float eps, void* workspace, void* stream);
int ucfvit_instnorm_cl_apply(const void* x, const void* res, void* y, const float* mean, const float* rstd, int64_t B, int64_t S, int64_t C,
                             float slope, void* stream); /* the apply pass alone, statistics given */
int ucfvit_instnorm_cl_apply2(const void* x, const float* mean, const float* rstd, const void* x2, const float* mean2, const float* rstd2, void* y,
                              int64_t B, int64_t S, int64_t C, float slope, void* stream);
int64_t ucfvit_instnorm_cl_bwd2_workspace(int64_t B, int64_t S, int64_t C);
int ucfvit_instnorm_cl_bwd2(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const void* x2, const float* mean2,
                            const float* rstd2, void* dx, void* dx2, int64_t B, int64_t S, int64_t C, int64_t ld_dy, float slope, void* workspace,
                            void* stream);

/* ------------------------------------------------------------------------------------------------------
 * UNETR convolutional decoder, convolutions.  Reference call sites: src/UCF_VIT/simple/arch.py:808-940 — monai's
 * blocks are chains of Conv3d(kernel 3, stride 1, padding 1, no bias) and ConvTranspose3d(kernel 2, stride 2, no bias); these entry
 * points replace torch.nn.functional.conv3d / conv_transpose3d (MIOpen) under them, and the 1x1x1 layers (residual projections, the
 * UnetOutBlock head, the transposed convolutions' channel mixing at small channel counts).
 * Activations are channels-last bf16 [B][X][Y][Z][C] (= torch.channels_last_3d memory of an [B, C, X, Y, Z] tensor).
 *
 * ucfvit_conv3d_fwd: y[v ldy + co] = bias[co] + sum_{tap, ci} w[co][ci][tap] x[v + tap - pad][ci] for co < cout_store (zero outside the
 *   volume; ksize 3 with padding 1, or ksize 1 = a pointwise layer over tall-skinny voxel rows), fp32 accumulation on MFMA, output bf16 or
 *   fp32 (out_dtype).  Cin in {8, 16, 32 k}, Cout = 16 k = the rows of w_packed; cout_store <= Cout channels are written with row stride
 *   ldy (a 4-class head writes [V][4] from a 16-row weight block); bias fp32 [Cout] or NULL; accumulate: y += result (the second of two
 *   data gradients that flow into the same input, e.g. a residual block's 3x3x3 and 1x1x1 branches).
 *   stats_partial (may be NULL): the instance-norm statistics of the OUTPUT as a by-product — per channel the count, the mean and
 *   M2 = sum (q - mean)^2 of the rounded outputs each wave stored (accumulated relative to a per-wave shift, so a channel whose |mean| is far
 *   larger than its spread keeps its variance bits), [B][rows][3][Cout] fp32 with rows = ucfvit_conv3d_fwd_stats_rows(...) (0: the kernel
 *   serving this shape has no such epilogue); ucfvit_instnorm_cl_stats_fold combines the rows (parallel-variance formula, double) into
 *   mean / rstd, which saves the statistics pass over the output.
 *   w_packed (bf16) holds the weights per 32-wide contraction step: with CPC = min(Cin, 32), TPS = 32 / CPC taps per step and
 *   NTS = ceil(ksize^3 / TPS) steps per channel chunk, w_packed[cc][ts][co][kk] = w[co][cc CPC + kk % CPC][tap] for tap = ts TPS + kk / CPC
 *   (zero when tap >= ksize^3); tap = (dx 3 + dy) 3 + dz.  The data gradient is the same call on dy with the flipped, transposed weights.
 * ucfvit_conv3d_wgrad: dw[tap][co][ci] = sum_v dy[v][co] x[v + tap - pad][ci] in fp32, written as
 *   [Cout / (16 MB)][Cin / CPC][ksize^3][16 MB][max(CPC, 16)] with MB = 2 when Cout % 32 == 0, else 1 (ucfvit_conv3d_wgrad_size floats; for
 *   CPC = 8 columns 8..15 are scratch).  workspace: ucfvit_conv3d_wgrad_workspace bytes (per-workgroup partials, folded in a fixed order:
 *   deterministic).
 * ucfvit_depth_to_space2: the transposed convolution is the GEMM x[V][Cin] * w[Cin][8 Cout] followed by this shuffle of
 *   cols [B Xi Yi Zi][(dx, dy, dz)][C] into out [B][2 Xi][2 Yi][2 Zi][C] (to_space = 1) — or its inverse for the backward pass (0).  The
 *   space tensor may be a channel slice of a wider channels-last buffer (voxel-row stride ld_space elements): the up-sampled map is written
 *   straight into the concatenation the next block reads; with `skip` (dense [..][Cs], to_space = 1) the skip connection is copied behind it
 *   in the same pass, so the concatenation is written as whole rows.
 * ucfvit_pad_channels8: fp32 N C D H W input [B][C][S] with C <= 8 -> bf16 channels-last [B][S][8] (channels C..7 zero): the input volume as
 *   an operand of the kernels above.
 * ucfvit_pad_rows8: V rows of C <= 8 values (src_dtype UCFVIT_F32 or UCFVIT_BF16, row stride ld elements) -> dense bf16 [V][8], columns C..7
 *   zero: the gradient of the output head's logits (_hip/conv.py:Conv1x1x1Fn.backward) as an operand of the kernels above, in one pass.
 * ------------------------------------------------------------------------------------------------------ */
int ucfvit_conv3d_fwd(const void* x, const void* w_packed, const float* bias, void* y, int64_t B, int64_t X, int64_t Y, int64_t Z, int64_t Cin,
                      int64_t Cout, int ksize, int64_t ldy, int64_t cout_store, int out_dtype, int accumulate, float* stats_partial, void* stream);
int64_t ucfvit_conv3d_fwd_stats_rows(int64_t B, int64_t X, int64_t Y, int64_t Z, int64_t Cin, int64_t Cout, int ksize, int has_bias);
int ucfvit_instnorm_cl_stats_fold(const float* partial, float* mean, float* rstd, int64_t B, int64_t S, int64_t C, int64_t rows, float eps,
                                  void* workspace /* B * 256 * 2 C floats, or NULL */, void* stream);
int64_t ucfvit_conv3d_wgrad_size(int64_t Cin, int64_t Cout, int ksize);
int64_t ucfvit_conv3d_wgrad_workspace(int64_t B, int64_t X, int64_t Y, int64_t Z, int64_t Cin, int64_t Cout, int ksize);
int ucfvit_conv3d_wgrad(const void* x, const void* dy, float* dw_packed, void* workspace, int64_t B, int64_t X, int64_t Y, int64_t Z,
                        int64_t Cin, int64_t Cout, int ksize, void* stream);
int ucfvit_depth_to_space2(const void* src, void* dst, int64_t B, int64_t Xi, int64_t Yi, int64_t Zi, int64_t C, int64_t ld_space, int to_space,
                           const void* skip, int64_t Cs, void* stream);
int ucfvit_pad_channels8(const float* src, void* dst, int64_t B, int64_t C, int64_t S, void* stream);
int ucfvit_pad_rows8(const void* src, int src_dtype, void* dst, int64_t V, int64_t C, int64_t ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UCFVIT_HIP_H */
