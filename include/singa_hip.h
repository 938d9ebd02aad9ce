/*
 * singa_hip.h — C ABI of libsinga_hip.so: the MI355X (gfx950) kernels behind the SINGA hot path.
 *
 * The reference (Isomorpfishm/SINGA) is pure Python; the seam it offers for this path is the set of tensor
 * operations inside model/EF_layers.py ("EF") and model/CProMG.py ("CP").  Each entry point below replaces one of
 * those call sites (file:line cited) and is what a ctypes binding on the reference side would call
 * (INTEGRATION.md shows the stub).  Conventions:
 *   - every pointer is a DEVICE pointer to caller-owned, contiguous memory unless a leading dimension is given;
 *     fp32 data, int32 indices; pointers are borrowed for the duration of the enqueue only;
 *   - functions only ENQUEUE work on `stream` (a hipStream_t passed as void*); no allocation, no synchronisation,
 *     no global mutable state after singa_init();
 *   - return 0 on success, a negative SINGA_E* code for argument errors, a positive hipError_t for launch errors;
 *     singa_last_error_string() describes the last failure of the calling thread;
 *   - edges of one edge type are sorted by destination node: row_ptr[n]..row_ptr[n+1] are the edges into node n.
 *
 * Coefficient orderings: node tensors are [N, K=(L+1)^2, C] l-primary; edge tensors between the
 * two rotations are "m-primary": [(l,0) l=0..L] ++ for m=1..M: [(l,+m) l=m..L] ++ [(l,-m) l=m..L], KR rows.
 * Reduced Wigner rows Wr[E, WSZ]: for l=0..L the rows |m|<=min(l,M) of the l-th Wigner block, row-major
 * [2*min(l,M)+1][2l+1]  (WSZ = 35 / 115 / 235 for L = 2 / 4 / 6 at M = 2).
 */
#ifndef SINGA_HIP_H
#define SINGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SINGA_OK 0
#define SINGA_E_NULL (-1)        /* required pointer is null */
#define SINGA_E_LMAX (-2)        /* unsupported (lmax, mmax): built for lmax in {2,4,6}, mmax = 2; EF:2208-2209 */
#define SINGA_E_SHAPE (-3)       /* inconsistent or unsupported dimension */
#define SINGA_E_NOINIT (-4)      /* singa_init() has not been called on this device */

/* A strided view of `rows` coefficient rows of width `ch` floats per edge: row r of edge e starts at
 * ptr + e*ld + r*ch.  Up to three segments make up one m-primary edge tensor (m = 0 | +-1 | +-2 GEMM outputs). */
typedef struct {
    const float* ptr;
    int64_t ld;     /* floats between consecutive edges */
    int32_t rows;   /* coefficient rows in this segment */
} singa_seg_t;

typedef struct {
    float* ptr;
    int64_t ld;
    int32_t rows;
} singa_seg_mut_t;

int singa_version(void);
const char* singa_last_error_string(void);

/* Upload the J matrices (reference model/Jd.pt, EF:2195-2198) for l = 0..lmax_max (<= 11) to the current device.
 * jd_flat: HOST pointer, blocks (2l+1)x(2l+1) row-major, concatenated. */
int singa_init(const double* jd_flat, int lmax_max);

/* Sizes the host needs to allocate buffers: KR, WSZ, RAD_ROWS for (lmax, mmax). */
int singa_dims(int lmax, int mmax, int* kr, int* wsz, int* rad_rows);

/* k1 - init_edge_rot_mat (EF:2286-2351): vec[E,3] edge vectors, rnd[E,3] the uniform [0,1) draws the reference takes
 * from torch.rand_like (Q6) -> rot[E,3,3] edge frames (rows z, x, -y; R x_hat = +y).  The reference's two guards
 * (EF:2292-2297 shortest edge, EF:2329 helper vector aligned with an edge) are reported through stats[2]:
 * stats[0] = min(stats[0], min |vec|), stats[1] = max(stats[1], max |cos(edge, helper)|) with a NaN propagated; the
 * caller initialises (+inf, 0) and judges the values (warning / error) after a read-back. */
int singa_edge_frames(const float* vec, const float* rnd, float* rot, float* stats, int E, void* stream);

/* k2 — SO3_Rotation.set_wigner / RotationToWignerDMatrix / wigner_D (EF:485-528, 2207-2229):
 * rot[E,3,3] edge frames -> Wr[E,WSZ] reduced Wigner rows. */
int singa_wigner_rows(const float* rot, float* wr, int E, int lmax, int mmax, void* stream);

/* k3-k6 — _expand_edge x2 + cat + SO3_Rotation.rotate + _m_primary + radial multiply (EF:326-328,1116-1119,494-497,
 * 354-355,822-824,847-850).  x_src[Ns,K,C], x_dst[Nd,K,C]; out[E,KR,2C] m-primary, already multiplied by
 * rad[E,RAD_ROWS,2C] (pass rad = NULL for no multiply). */
int singa_gather_rotate_fwd(const float* x_src, const float* x_dst, const int32_t* src, const int32_t* dst,
                            const float* wr, const float* rad, float* out, int E, int C, int lmax, int mmax,
                            void* stream);
/* backward: g_out[E,KR,2C] -> g_rad[E,RAD_ROWS,2C] (may be NULL), gx_dst[Nd,K,C] (segmented over row_ptr),
 * gx_src[Ns,K,C] (segmented over col_ptr / eperm = edge ids sorted by source). */
int singa_gather_rotate_bwd(const float* g_out, const float* x_src, const float* x_dst, const int32_t* src,
                            const int32_t* dst, const float* wr, const float* rad, const int32_t* row_ptr,
                            const int32_t* col_ptr, const int32_t* eperm, float* g_rad, float* gx_src, float* gx_dst,
                            int E, int Ns, int Nd, int C, int lmax, int mmax, void* stream);

/* k10 / k13 — alpha scaling + _l_primary + SO3_Rotation.rotate_inv (with rescale) + _reduce_edge (index_add_)
 * (EF:1186-1199, 499-505, 342-351; EdgeDegreeEmbedding EF:116-147 when m0_only != 0).
 * msg: m-primary edge tensor with CH channels in `nseg` segments; alpha[E,heads] or NULL; out[Nd,K,CH].
 * Channel c belongs to head c / (CH/heads).  out is multiplied by out_scale (1/23.395... for k13). */
int singa_rotate_back_scatter_fwd(const singa_seg_t* msg, int nseg, const float* alpha, const float* wr,
                                  const int32_t* row_ptr, float* out, int Nd, int CH, int heads, int lmax, int mmax,
                                  int m0_only, float out_scale, void* stream);
/* backward: g_out[Nd,K,CH] -> g_msg (same segment layout as msg, contiguous per segment), g_alpha_part[E,CH]
 * (per-channel partial of d/d alpha; the caller sums the CH/heads channels of each head; NULL if alpha is NULL). */
int singa_rotate_back_scatter_bwd(const float* g_out, const singa_seg_t* msg, const singa_seg_mut_t* g_msg, int nseg,
                                  const float* alpha, const float* wr, const int32_t* row_ptr, float* g_alpha_part,
                                  int Nd, int CH, int heads, int lmax, int mmax, int m0_only, float out_scale,
                                  void* stream);

/* k9a - attention logits (EF:1175-1178): x0_alpha.view(E, heads, A) -> LayerNorm(A) -> SmoothLeakyReLU(0.2) -> dot with
 * alpha_dot[heads, A].  h0: the m = 0 SO(2)-conv output whose first heads*A columns are the alpha inputs (row stride ld).
 * backward: g_x[E, heads*A]; part[singa_alpha_logits_nslots(E), (2+heads)*A] = per-slot partials of (d ln_w, d ln_b,
 * d alpha_dot) which the caller reduces with singa_colsum.  heads = 7, A = 32. */
int singa_alpha_logits_nslots(int E);
int singa_alpha_logits_fwd(const float* h0, long long ld, const float* ln_w, const float* ln_b, const float* dot, float* logits,
                           int E, int heads, int A, float eps, void* stream);
int singa_alpha_logits_bwd(const float* h0, long long ld, const float* ln_w, const float* ln_b, const float* dot,
                           const float* g_logits, float* g_x, float* part, int E, int heads, int A, float eps, void* stream);
/* the same with a row stride for g_x (>= heads * A): the gradient lands in a column block of a wider tensor */
int singa_alpha_logits_bwd_ld(const float* h0, long long ld, const float* ln_w, const float* ln_b, const float* dot,
                              const float* g_logits, float* g_x, long long ld_gx, float* part, int E, int heads, int A, float eps,
                              void* stream);

/* k9 (softmax part) — torch_geometric.utils.softmax / torch_scatter.scatter_softmax over destination segments
 * (EF:1180; CP:66): out = exp(x - segmax) / (segsum + eps).  x, out: [E, H].  dense_segments != 0 (and H == 4) selects the
 * wavefront-per-segment variant for segments of tens of edges (the kNN graphs of CP:293-298); 0 = thread per (segment, head),
 * right for the ~8-edge segments of the bonded graphs. */
int singa_segment_softmax_fwd(const float* x, const int32_t* row_ptr, float* out, int N, int H, float eps,
                              int dense_segments, void* stream);
int singa_segment_softmax_bwd(const float* y, const float* gy, const int32_t* row_ptr, float* gx, int N, int H,
                              int dense_segments, void* stream);

/* k15 — alpha-weighted scatter_sum of per-edge messages (CP:71-74): out[N,H,F] = sum_e w[e,H] * v[e,H,F]. */
int singa_segment_wsum_fwd(const float* w, const float* v, const int32_t* row_ptr, float* out, int N, int H, int F,
                           void* stream);
int singa_segment_wsum_bwd(const float* g_out, const float* w, const float* v, const int32_t* row_ptr, float* gw,
                           float* gv, int N, int H, int F, void* stream);

/* k15b — CProMG MultiHeadAttention (CP:59-74) with weight_k_lin / weight_v_lin hoisted to node level by linearity, so
 * no [E,H,D] tensor exists.  Edges sorted by row (centre node): row_ptr[N+1], col[E]; col_ptr/eperm = edges grouped by
 * col (for the gradient of the gathered operand); row[E] = centre of each edge.  H = 4, D = 32, F = 64.
 *   qk[e,h]    = scale * sum_d qp[row,h,d] wk[e,d] hk[col,h,d] + cterm[row,h]      (CP:61-65)
 *   out[n,h,f] = sum_e alpha[e,h] wv[e,f] hv[col,h,f]                                (CP:70-74) */
int singa_edge_logits_fwd(const float* qp, const float* wk, const float* hk, const float* cterm, const int32_t* row_ptr,
                          const int32_t* col, float* qk, int N, int H, int D, float scale, void* stream);
int singa_edge_logits_bwd(const float* g, const float* qp, const float* wk, const float* hk, const int32_t* row_ptr,
                          const int32_t* col, const int32_t* col_ptr, const int32_t* eperm, const int32_t* row,
                          float* g_qp, float* g_wk, float* g_hk, float* g_cterm, int N, int H, int D, float scale,
                          void* stream);
int singa_gather_wsum_fwd(const float* alpha, const float* wv, const float* hv, const int32_t* row_ptr, const int32_t* col,
                          float* out, int N, int H, int F, void* stream);
int singa_gather_wsum_bwd(const float* g, const float* alpha, const float* wv, const float* hv, const int32_t* row_ptr,
                          const int32_t* col, const int32_t* col_ptr, const int32_t* eperm, const int32_t* row,
                          float* g_alpha, float* g_wv, float* g_hv, int N, int H, int F, void* stream);

/* k15d — y[M,n] = softplus(u + b) - ln 2 and gu = g * sigmoid(u + b): `ShiftedSoftplus` between the two Linears of
 * `weight_k_net` / `weight_v_net` (reference model/CProMG.py:33-48) with the first Linear's bias folded in (the GEMM then
 * needs no bias epilogue; d b = colsum(gu)).  n a multiple of 4, rows contiguous. */
int singa_bias_ssp_fwd(const float* u, const float* b, float* y, long long M, int n, void* stream);
int singa_bias_ssp_bwd(const float* u, const float* b, const float* g, float* gu, long long M, int n, void* stream);

/* k16 — y = LayerNorm(a + r) over rows of C = 256 channels, r optional (NULL): the residual LayerNorms of the CProMG
 * transformer (reference model/CProMG.py:78, 105, 158, 176, 191, 264; torch.nn.LayerNorm: biased variance, eps inside the
 * root).  Backward: gs[M,C] = gradient w.r.t. the sum a + r (the gradient of both), part[singa_ln256_nparts(M)][2C] =
 * per-wavefront partial sums [d gamma | d beta] to be reduced with singa_colsum. */
int singa_ln256_nparts(long long M);
int singa_ln256_fwd(const float* a, const float* r, const float* gamma, const float* beta, float* y, long long M, int C,
                    float eps, void* stream);
int singa_ln256_bwd(const float* a, const float* r, const float* gamma, const float* g, float* gs, float* part, long long M,
                    int C, float eps, void* stream);

/* k17 — one new position of a CProMG decoder layer for every live row of a beam search (reference model/CProMG.py:134-191,
 * 346-383 evaluated incrementally; the reference's model/BeamSearch.py:82 re-runs the whole decoder on the prefix).  Rows R =
 * proteins x beams, one workgroup per row.  All weight matrices are passed TRANSPOSED, wT[in][out].  Built for hidden 256,
 * 4 heads, 32 key / 64 value channels per head, FFN 1024.
 *   self-attention: wqkv_t[256][512] = [W_Q | W_K | W_V]^T, k_cache[R][4][P][32], v_cache[R][4][P][64]; the new key / value
 *     are written at *pos (device scalar) and the row attends positions 0..*pos; y = LN(W_O ctx + b_O + x).  P <= 256.
 *   cross-attention: ck[B][4][32][S] / cv[B][4][S][64] = encoder keys (transposed) / values projected once per protein,
 *     pad[B][S] != 0 = padding (score -1e9), row r uses protein r / beams; z = LN(W_O ctx + b_O + y).  S <= 1024.
 *   ffn: out = LN(W_2 relu(W_1 z + b_1) + b_2 + z), w1_t[256][1024], w2_t[1024][256]. */
int singa_dec_self_attn(const float* x, const float* wqkv_t, const float* bqkv, const float* wo_t, const float* bo,
                        const float* gamma, const float* beta, float* k_cache, float* v_cache, const long long* pos, int R, int P,
                        float* y, float eps, void* stream);
int singa_dec_cross_attn(const float* y, const float* wq_t, const float* bq, const float* ck, const float* cv,
                         const unsigned char* pad, const float* wo_t, const float* bo, const float* gamma, const float* beta,
                         int R, int beams, int S, float* z, float eps, void* stream);
int singa_dec_ffn(const float* z, const float* w1_t, const float* b1, const float* w2_t, const float* b2, const float* gamma,
                  const float* beta, int R, float* out, float eps, void* stream);

/* k15c — the two per-edge MLPs of the CProMG graph attention (reference model/CProMG.py:41-48 `weight_k_net`,
 * `weight_v_net` = Linear -> ShiftedSoftplus -> Linear, applied at CP:58 and CP:68): wk[E,HK] and wv[E,HV] from
 * attr[E,CIN] in one pass on the f32 MFMA, hidden activations never stored.  Weight matrices in nn.Linear's own layout:
 * w1t* = weight[H][CIN] of the first Linear, w2t* = weight[H][H] of the second; b1*, b2* [H].  Built for CIN = 64, HK = 32, HV = 64
 * (config model.encoder: edge_channels 64, key_channels 128 / 4 heads, hidden_channels 256 / 4 heads). */
int singa_edge_mlp_fwd(const float* attr, const float* w1tk, const float* b1k, const float* w2tk, const float* b2k,
                       const float* w1tv, const float* b1v, const float* w2tv, const float* b2v, float* wk, float* wv, int E,
                       int CIN, int HK, int HV, void* stream);

/* backward of ONE of the nets of k15c (H hidden = H output units, 32 or 64): recomputes the hidden units and accumulates
 * all four parameter gradients on the MFMA.  The hidden units are handled in slices of 32 (H/32 workgroups per edge range);
 * part[singa_edge_mlp_bwd_nparts(E, H)][H*64 + H + H*H + H] holds, per edge range, the partial
 * [dW1 [H][64] | db1 [H] | dW2 [H][H] | db2 [H]] in the parameters' own layouts (nn.Linear weights [out][in]), to be reduced
 * over the edge ranges with singa_colsum / singa_colsum_multi.
 * w1t = weight[H][CIN] of the first Linear, w2 = weight[H][H] of the second (nn.Linear's own layouts). */
int singa_edge_mlp_bwd_nparts(int E, int H);
int singa_edge_mlp_bwd(const float* attr, const float* g_out, const float* w1t, const float* b1, const float* w2, float* part,
                       int E, int CIN, int H, void* stream);

/* k18 — softmax(mask ? -1e9 : scale * s) over the last axis of the attention scores s[BH, T, S] (reference model/CProMG.py:
 * 111-115, 140-146: `/ np.sqrt(d_k)`, `masked_fill_(attn_mask, -1e9)`, `Softmax(dim=-1)`), and its gradient
 * gs = scale * p * (gp - sum_j gp_j p_j), zero at masked positions.  mask: bytes [B, T, S] addressed as
 * mask[b * mask_stride_b + t * mask_stride_t + j] (an expanded padding mask has mask_stride_t = 0); row bh uses b = bh / heads. */
int singa_masked_softmax_fwd(const float* s, const unsigned char* mask, long long mask_stride_b, long long mask_stride_t, float* p,
                             int BH, int T, int S, int heads, float scale, void* stream);
int singa_masked_softmax_bwd(const float* p, const float* gp, const unsigned char* mask, long long mask_stride_b,
                             long long mask_stride_t, float* gs, int BH, int T, int S, int heads, float scale, void* stream);

/* k19 — dense multi-head attention core of the CProMG decoder / cross attentions (reference model/CProMG.py:107-117, 136-148:
 * scores = Q K^T / sqrt(d_k), masked_fill_(mask, -1e9), softmax, context = attn V) on the f32 MFMA, flash-style: q[BH,T,DK],
 * k[BH,S,DK], v[BH,S,DV] contiguous, mask bytes addressed as in k18, ctx[BH,T,DV], lse[BH,T,2] = (maximum, 1 / sum of
 * exponentials) of each score row (kept for the backward).  Built for DK = 32, DV = 64.
 * token_major != 0: q / k / v / ctx (and the gradients of _bwd) are [B, T|S, heads, D] - the layout in which the W_Q / W_K /
 * W_V projections produce them and `linear` consumes the context (model/CProMG.py:96-117) - instead of [B*heads, T|S, D]:
 * the reference's `.view(B, -1, heads, d).transpose(1, 2)` copies and the `.transpose(1, 2).contiguous()` of the context
 * disappear.  lse and dsum stay [B*heads, T].  ld_q / ld_k / ld_v (token_major only; 0 = dense): floats between consecutive
 * tokens of q / k / v - and of g_q / g_k / g_v in _bwd - when they are column blocks of ONE fused projection output
 * [B, T, 128 + 128 + 256] (W_Q | W_K | W_V evaluated as one GEMM) instead of three dense tensors; ctx / g_ctx stay dense. */
int singa_attn_fwd(const float* q, const float* k, const float* v, const unsigned char* mask, long long mask_stride_b,
                   long long mask_stride_t, float* ctx, float* lse, int BH, int T, int S, int heads, int DK, int DV,
                   int token_major, long long ld_q, long long ld_k, long long ld_v, float scale, void* stream);
/* gradients g_q, g_k, g_v from g_ctx: scores are recomputed from q, k and lse (nothing of size T x S is ever stored);
 * dsum[BH,T] is scratch (rowsum(g_ctx * ctx), written by the first pass and read by the second). */
int singa_attn_bwd(const float* q, const float* k, const float* v, const unsigned char* mask, long long mask_stride_b,
                   long long mask_stride_t, const float* ctx, const float* lse, const float* g_ctx, float* g_q, float* g_k,
                   float* g_v, float* dsum, int BH, int T, int S, int heads, int DK, int DV, int token_major, long long ld_q,
                   long long ld_k, long long ld_v, float scale, void* stream);

/* k6a — LayerNorm over C = 16 channels followed by SiLU: the `nn.LayerNorm`, `nn.SiLU` pair inside RadialFunction
 * (reference model/EF_layers.py:1634-1657, net.1/net.2 and net.4/net.5).  x, out, g_out, g_x: [M, C] contiguous; biased
 * variance, eps inside the root (torch.nn.LayerNorm).  The backward recomputes the statistics and writes per-thread
 * partial sums part[singa_ln_silu_nparts(M)][2C] = [d gamma | d beta], to be reduced with singa_colsum. */
int singa_ln_silu_nparts(long long M);
int singa_ln_silu_fwd(const float* x, const float* gamma, const float* beta, float* out, long long M, int C, float eps,
                      void* stream);
int singa_ln_silu_bwd(const float* x, const float* gamma, const float* beta, const float* g_out, float* g_x, float* part,
                      long long M, int C, float eps, void* stream);

/* k8 — SeparableS2Activation (EF:1736-1773): rows -> S2 grid (to_grid[G,KIN]) -> SiLU -> rows (from_grid[G,KIN]);
 * row 0 of the output is SiLU(gate).  x: KIN rows of C channels in `nseg` segments; gate[E, ldg]; out[E,KIN,C].
 * Grid matrices are given in the row order of x (the host permutes them for m-primary inputs). */
int singa_s2act_fwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* to_grid,
                    const float* from_grid, float* out, int E, int C, int KIN, int G, void* stream);
int singa_s2act_bwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* to_grid,
                    const float* from_grid, const float* g_out, float* gx, float* g_gate, int E, int C, int KIN,
                    int G, void* stream);

/* k8, separable form (what the product calls): to_grid[(b,a),i] = P[b,i] * A[a,mc(i)], from_grid = Q[b,i] * A[a,mc(i)]
 * (singa_amd.so3.s2_grid_factors): Legendre transform per beta ring, Fourier transform per ring - ~3x fewer FMAs.
 * nseg = 3, C = 128: attention grid [lmax][2] on m-primary rows; nseg = 1, C = 512: FFN grid [lmax][lmax] on [K, C]. */
int singa_s2act_sep_fwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* P, const float* Q,
                        const float* A, float* out, int E, int C, int lmax, void* stream);
int singa_s2act_sep_bwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* P, const float* Q,
                        const float* A, const float* g_out, float* gx, float* g_gate, int E, int C, int lmax,
                        void* stream);
/* The feed-forward block's backward pass through [SeparableS2Activation -> SO3_LinearV2(512 -> 16)] (EF:256-262) in ONE pass over
 * the hidden tensor: the gradient at the activation's output is formed on the fly from the small gradient g_small [N, K, 16] of
 * the linear's output and its weight W2 [L+1][16][512] (gy[n, i, c] = sum_u g_small[n, i, u] W2[l(i)][u][c]) instead of being
 * written by an expand launch and read back.  x: the activation's saved input [N, K, 512]; gate [N, 512] (row stride ldg);
 * P, Q: the node grid's Legendre tables (singa_s2act_sep_fwd); outputs gx [N, K, 512], g_gate [N, 512]. */
int singa_s2act_ffn_bwd(const float* x, const float* gate, int64_t ldg, const float* P, const float* Q, const float* g_small,
                        const float* W2, float* gx, float* g_gate, int N, int C, int lmax, void* stream);
/* the same with the input gradient written into `nseg` segments that mirror x's (rows equal, ld >= rows * C) and a row
 * stride for g_gate (>= C): model/EF_layers.py:1148-1178 feeds this activation from column blocks of the three outputs of an
 * SO(2) convolution, and the gradients of those outputs are assembled in place (no concatenation pass). */
int singa_s2act_sep_bwd_seg(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* P, const float* Q,
                            const float* A, const float* g_out, const singa_seg_mut_t* gx, float* g_gate, int64_t ld_gg, int E,
                            int C, int lmax, void* stream);

/* k12 — EquivariantRMSNormArraySphericalHarmonicsV2 as instantiated by get_normalization_layer (EF:2155-2192, 2273):
 * x[N,K,C] -> y[N,K,C]; weight[L+1,C], bias[C]. */
int singa_so3_rmsnorm_fwd(const float* x, const float* weight, const float* bias, float* y, int N, int C, int lmax,
                          float eps, void* stream);
/* backward: gx[N,K,C]; gw_part[nparts,L+1,C] (already summed over the 2l+1 rows of each degree) and gb_part[nparts,C] are
 * per-wave partial sums the caller reduces (nparts = singa_so3_rmsnorm_nparts(N)). */
int singa_so3_rmsnorm_nparts(int N);
int singa_so3_rmsnorm_bwd(const float* x, const float* weight, const float* gy, float* gx, float* gw_part,
                          float* gb_part, int N, int C, int lmax, float eps, void* stream);
/* the same with gx = (norm's input gradient) + g_add: the gradient of the residual branch that leaves the norm's input
 * (x + f(norm(x)), EF:1383-1384, 1405-1406) is added here instead of by a separate pass over both tensors */
int singa_so3_rmsnorm_bwd_add(const float* x, const float* weight, const float* gy, const float* g_add, float* gx, float* gw_part,
                              float* gb_part, int N, int C, int lmax, float eps, void* stream);

/* Column sums out[n] = sum_i x[i*ld + j] (bias and broadcast gradients; replaces torch's multi-block `sum(0)`, which
 * is not replay-safe on this ROCm build).  work: singa_colsum_work(M, n) floats. */
long long singa_colsum_work(long long M, int n);
int singa_colsum(const float* x, long long ld, long long M, int n, float* work, float* out, void* stream);

/* The parameter-gradient reductions of one backward pass in two launches per 36 jobs (replaces one torch `sum(0)` /
 * AccumulateGrad pair per parameter: torch/autograd, used by every bias / affine / split-GEMM weight gradient of
 * model/EF_layers.py and model/CProMG.py).  Job k adds the column sums of x[k] [M[k], n[k]] (row stride ld[k]) INTO its
 * destination segments: segments job_seg0[k] .. job_seg0[k+1]-1 (the last job ends at n_segs); segment q receives columns
 * seg_col0[q] .. (next segment's col0 or n[k]) - 1 at seg_dst[q][column - seg_col0[q]].  The first segment of a job
 * starts at column 0, col0 ascends.  The tables are host memory and are consumed by the call (they ride in the kernel
 * arguments, so a captured launch needs no copy).  work: sum over jobs of singa_colsum_multi_work(M, n) floats. */
long long singa_colsum_multi_work(long long M, int n);
int singa_colsum_multi(int n_jobs, const float* const* x, const long long* ld, const long long* M, const int* n,
                       const int* job_seg0, int n_segs, const int* seg_col0, float* const* seg_dst, float* work,
                       long long work_floats, void* stream);

/* Adam step of train.py:127 (torch.optim.Adam: lr, betas, eps; no weight decay) for ALL parameter tensors in one launch.
 * p/g/m/v: DEVICE arrays of device pointers (one per tensor); sizes[t] = elements of tensor t; (chunk_tensor, chunk_off)
 * [nchunks]: the flattened work list, `chunk` elements each; step (float, number of steps taken) and lr live in device
 * memory so that the launch can be replayed from a HIP graph. */
int singa_adam_step(float* const* p, const float* const* g, float* const* m, float* const* v, const long long* sizes,
                    const int32_t* chunk_tensor, const long long* chunk_off, int nchunks, int chunk, float* step,
                    const float* lr, float beta1, float beta2, float eps, void* stream);

/* k7 / k11 - f32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32) for the per-edge contractions of the SO(2)
 * convolutions (reference model/EF_layers.py:715-729, 782, 807-875: fc_m0 and the so2_m_conv[m].fc Linears applied to
 * m-primary edge rows) and the per-degree SO3_LinearV2 (EF:655-671).  Up to SINGA_GEMM_MAX independent problems per
 * launch:  c[i, j] = sum_r A(i, r) * B(r, j) (+ bias[j]).
 *   a_r_contig != 0: A is stored [I rows][R] with row pitch lda; else A is stored [R rows][I].
 *   b_r_contig != 0: B is stored [J][R] (an nn.Linear weight [out, in]); else [R rows][J].
 *   Rows of A, of an [R][J] B and of C may be grouped: row i lives at (i / group) * group_ld + (i % group) * ld (the
 *   2l+1 coefficient rows of degree l inside [N, K, C] node tensors); group = 0: plain rows.
 *   Built combinations: (1,1) y = x W^T; (1,0) dx = dy W; (0,0) dW = dy^T x.
 *   splits > 1: the reduction range is cut into `splits` chunks; chunk s writes a dense [I, J] slab (ldc = J, no bias, no
 *   grouping) at c + s * c_split_stride, which the caller adds up (singa_colsum).
 *   All problems of a launch use 128 x 128 output tiles, 128 x 32 when every J <= 32, 32 x 128 when every I <= 32, or
 *   64 x 64 when the launch has fewer than 384 tiles of 128 x 128.
 * Contiguous axes (of A, B and of the result: J, ldc) must be multiples of 4 floats and 16-byte aligned.  Enqueue-only
 * on `stream`. */
#define SINGA_GEMM_MAX 12
typedef struct {
    const float* a;
    const float* b;
    float* c;
    const float* bias;        /* [J] or NULL */
    int64_t lda, ldb, ldc;
    int32_t I, J, R;
    int32_t a_group, b_group, c_group;           /* rows per group (0 = plain rows) */
    int64_t a_group_ld, b_group_ld, c_group_ld;
    int64_t c_split_stride;                       /* floats between the partial slabs of consecutive splits */
    const float* mask;        /* NULL, or a tensor with c's (plain-row) layout: c is zeroed where mask <= 0 (ReLU gradient) */
    const float* addend;      /* NULL, or a tensor with c's layout (same pitches / grouping) that is added to the product (a sum of two
                                 Linears' outputs, CP:77,400: `centroid_lin(x) + aggr_msg`; the residual behind an SO3_LinearV2,
                                 EF:1383-1384, 1405-1406) */
    int32_t relu;             /* != 0: c = max(c, 0) after bias and addend (nn.ReLU behind a Linear, CP:171,188) */
    float* asum;              /* NULL, or ((0, 0) form only) asum[s * asum_stride + i] = sum over split s's reduction rows of A(r, i):
                                 the bias gradient sum_m dy[m, n] computed while dW = dy^T x streams dy (no separate pass) */
    int64_t asum_stride;      /* floats between the splits' rows of asum (>= I) */
} singa_gemm_t;
int singa_gemm_f32(const singa_gemm_t* probs, int n, int a_r_contig, int b_r_contig, int splits, void* stream);

/* k7c - the order-m > 0 blocks of an SO(2) convolution (reference model/EF_layers.py:677-729: `x = self.fc(x)` on the +m and -m
 * rows, then out_r = x_r[:, 0] - x_i[:, 1], out_i = x_r[:, 1] + x_i[:, 0]) as COMPLEX products with three real multiplications
 * instead of four (3M): for every problem
 *     C_re(i, j) = sum_r a(i, r) c(r, j) - b(i, r) sigma d(r, j),    C_im(i, j) = sum_r a(i, r) sigma d(r, j) + b(i, r) c(r, j)
 * where (a, b) are the real and imaginary part of the A operand, (c, d) of the B operand; the imaginary part of each operand
 * and of the result lies a_im / b_im / c_im ELEMENTS behind the real part (same pitches).  Operand forms as singa_gemm_f32:
 * (1, 1) forward (A = [x_+ | x_-] rows, a_im = K; B = fc.weight [2N, K], b_im = N * K; C = [out_r | out_i], c_im = N; sigma = +1),
 * (1, 0) d input (A = [g_r | g_i], B = fc.weight as [r = n][j = k], sigma = -1, C = [dx_+ | dx_-]),
 * (0, 0) d weight (A = [g_r | g_i] as [r = e][i = n], B = [x_+ | x_-] as [r = e][j = k], sigma = -1, C = d fc.weight [2N, K],
 * c_im = N * K; splits > 1: dense partial slabs c_split_stride apart).  I, J, R count COMPLEX rows / columns / reduction
 * indices.  Exact f32 products and accumulation; the three-product form rounds differently from the four-product one (a few
 * 1e-7 of the operands' magnitudes). */
#define SINGA_CGEMM_MAX 4
typedef struct {
    const float* a;
    const float* b;
    float* c;
    int64_t lda, ldb, ldc;
    int64_t a_im, b_im, c_im;
    int64_t c_split_stride;
    int I, J, R;
    float sigma;
} singa_cgemm_t;
int singa_cgemm3m_f32(const singa_cgemm_t* probs, int n, int a_r_contig, int b_r_contig, int splits, void* stream);

/* k11s - SO3_LinearV2 (model/EF_layers.py:655-671) between 16 and C = 512 channels (the feed-forward block, EF:232-262) or
 * C = 112 (the attention's output projection, EF:1201-1204; there only the backward maps 16 -> 112 channels), where
 * the contraction is only 16 long: VALU kernels, thread = one of the 512 channels, whole 2 KB rows of the big tensor per access.
 * _expand: big[N, K, 512] = sum_u small[N, K, 16][.., u] * W[l(k)][c][u] (+ bias[c] on the l = 0 row; bias may be NULL), W
 * addressed as W[l * w_l + c * w_c + u * w_u] (weight[l][c][u]: w_c = 16, w_u = 1; weight[l][u][c]: w_c = 1, w_u = 512).
 * _reduce: partial weight gradients part[singa_so3_skinny_nparts(N, lmax, C)][(L+1) * 16 * C (+ C if bias_row)], each row
 * [l][c][u] (out_cu != 0) or [l][u][c], = sum over the part's nodes and the rows k of degree l of small[n, k, u] * big[n, k, c];
 * the bias row is sum_n big[n, 0, c].  The caller adds the parts up (singa_colsum / singa_colsum_multi). */
int singa_so3_skinny_nparts(int N, int lmax, int C);
int singa_so3_skinny_expand(const float* small, const float* W, long long w_l, long long w_c, long long w_u, const float* bias,
                            float* big, int N, int C, int lmax, void* stream);
int singa_so3_skinny_reduce(const float* small, const float* big, float* part, int N, int C, int lmax, int out_cu, int bias_row,
                            void* stream);

/* SO2_m_Convolution (reference model/EF_layers.py:677-729): its Linear fc (weight w [2h, k], Wr = w[:h], Wi = w[h:]) applied to
 * the +m rows x_+ and the -m rows x_- with the recombination out_r = fc_r(x_+) - fc_i(x_-), out_i = fc_r(x_-) + fc_i(x_+)
 * (EF:721-729) equals ONE Linear with the block weight out[2h, 2k] = [[Wr, -Wi], [Wi, Wr]] on [x_+ | x_-].  _fwd builds it;
 * _bwd maps the block weight's gradient back: g_w[:h] = G[:h, :k] + G[h:, k:], g_w[h:] = G[h:, :k] - G[:h, k:] (accumulate != 0:
 * added to g_w). */
int singa_block_weight_fwd(const float* w, float* out, int h, int k, void* stream);
int singa_block_weight_bwd(const float* g_block, float* g_w, int h, int k, int accumulate, void* stream);

/* out[m] = scale * sum_d x[m, d] * b[d], D = 32: the bias term q . b of the graph attention's logits when `weight_k_lin` is
 * hoisted to the query side (reference model/CProMG.py:61-65: q . (W (w * k) + b) = (q W) . (w * k) + q . b).  _bwd: gx[m, d] =
 * scale g[m] b[d] and part[singa_rowdot_nparts(M)][32] = per-workgroup partial sums of scale g[m] x[m, d] (the caller adds them
 * up: singa_colsum / singa_colsum_multi). */
int singa_rowdot_nparts(long long M);
int singa_rowdot_fwd(const float* x, const float* b, float* out, long long M, int D, float scale, void* stream);
int singa_rowdot_bwd(const float* g, const float* x, const float* b, float* gx, float* part, long long M, int D, float scale,
                     void* stream);

/* n2 - Laplacian positional encoding (reference model/CProMG.py:562-571 `lap_pe` -> dgl.lap_pe(g, k), called inside forward
 * at model/GAN.py:71,77): for each of B graphs the kout (<= 8) eigenvectors after the smallest of its normalised Laplacian
 * I - D^-1/2 A D^-1/2 (A[src, dst] = 1 for every edge, repeats count once; D = in-degree clipped at 1; symmetrised), entry
 * of largest magnitude made positive, written as fp32 rows out[first[b] + i, 0..kout-1]; graphs with fewer than kout + 1
 * atoms get zero columns.  Edges: esrc / edst are LOCAL atom indices (0 .. nnodes[b] - 1), grouped by graph, graph b owning
 * eptr[b] .. eptr[b+1] - 1.  A: [B, ld, ld] doubles of scratch (ld >= max nnodes, <= 896; not initialised by the caller),
 * work: singa_lap_pe_work(B, ld) doubles.  One workgroup per graph; all O(n^3) work is per connected component (Householder
 * tridiagonalisation, Sturm multi-section, inverse iteration, back-transformation). */
int singa_lap_pe_work(int B, int ld);
int singa_lap_pe(double* A, const int32_t* esrc, const int32_t* edst, const int32_t* eptr, const int32_t* nnodes, const int32_t* first,
                 double* work, float* out, int B, int ld, int kout, void* stream);

/* Total 2-norm of all gradients over the same (tensor, chunk) table as singa_adam_step: torch.nn.utils.clip_grad_norm_'s
 * norm (reference train.py:126), deterministic and HIP-graph replayable.  partial: nchunks floats of scratch; out: 1 float. */
int singa_grad_norm(const float* const* g, const long long* sizes, const int32_t* chunk_tensor, const long long* chunk_off,
                    int nchunks, int chunk, float* partial, float* out, void* stream);

/* n1 - kNN graphs of the CProMG encoders: torch_cluster.knn_graph(pos, k, batch, flow='target_to_source') at reference
 * model/CProMG.py:293 (k = 48, protein atoms) and :330 (k = 30, ligand atoms).  pos [N, 3]; batch [N] molecule of every atom (ids outside
 * [0, B) = atoms of no molecule: no neighbours); ptr [B + 1] first atom of every molecule (the atoms of a molecule are contiguous, as
 * PyG's collate leaves them); max_nodes >= the largest molecule (<= 2048).  row / col [N * k] int64: row = centre atom, col = its
 * neighbours in order of increasing distance (exact fp32 coordinate differences, ties to the lower index); -1 in both where a slot does
 * not exist (molecules with fewer than k + 1 atoms). */
int singa_knn_graph(const float* pos, const int32_t* batch, const long long* ptr, int B, int N, int k, int max_nodes, long long* row,
                    long long* col, void* stream);

/* n1 - edge features of those kNN graphs in their final layout (reference model/CProMG.py:295-298: GaussianSmearing of the undirected
 * edge lengths, then get_laplacian with 2-D edge weights (Q12): off-diagonal entries -w, one appended self loop per node carrying the
 * sum of its row's weights).  len [n_real] lengths of the row-sorted undirected edges; ptr [N + 1] first edge of every centre node in
 * that list (edges >= n_real: the inert padding edges of a padded batch, zeros); offset [G] the Gaussians' centres, coeff = -0.5 /
 * spacing^2; out [ptr[N] + N, G]: row e + i = -exp(coeff (len[e] - offset)^2) for edge e of node i, row ptr[i+1] + i = the sum of node
 * i's rows (positive).  G = 64. */
int singa_knn_edge_attr(const float* len, const int32_t* ptr, long long n_real, const float* offset, float coeff, float* out, int N,
                        int G, void* stream);

/* Measurement helpers (bench.py): exact per-dispatch timing of the scatter-TP forward kernel with start/stop events
 * attached to the dispatch (hipExtLaunchKernelGGL) on the caller's stream, and a copy kernel with the segment kernels'
 * access shape for calibrating the PMC byte counters. */
int singa_prof_enable(int on);
int singa_prof_hint_edges(int E);
int singa_prof_collect(float* ms, int* edges, int* nodes, int cap); /* after synchronising; returns #records */
/* kernel tags of the profiled dispatches (bench.py's roofline block) */
#define SINGA_PROF_K10_FWD 1     /* rotate_back_scatter forward ("scatter-TP") */
#define SINGA_PROF_K10_BWD 2
#define SINGA_PROF_K4_FWD 3      /* gather_rotate forward */
#define SINGA_PROF_K4_BWD_RAD 4  /* gather_rotate backward w.r.t. the radial weights (edge-parallel) */
#define SINGA_PROF_K4_BWD_DST 5  /* ... w.r.t. the destination node rows */
#define SINGA_PROF_K4_BWD_SRC 6  /* ... w.r.t. the source node rows */
#define SINGA_PROF_GEMM_NT 7     /* singa_gemm_f32 (1,1): nodes field = number of 128 x 128 tiles */
#define SINGA_PROF_GEMM_NN 8
#define SINGA_PROF_GEMM_TN 9
#define SINGA_PROF_S2_EDGE_FWD 10   /* separable S2 activation on the attention grid (edge rows, 128 channels) */
#define SINGA_PROF_S2_EDGE_BWD 11
#define SINGA_PROF_S2_NODE_FWD 12   /* ... on the feed-forward grid (node rows, 512 channels) */
#define SINGA_PROF_S2_NODE_BWD 13
#define SINGA_PROF_CGEMM_NT 14    /* singa_cgemm3m_f32 (1,1): nodes field = number of 128 x 64 complex tiles */
#define SINGA_PROF_CGEMM_NN 15
#define SINGA_PROF_CGEMM_TN 16
int singa_prof_collect_tagged(float* ms, int* tags, int* edges, int* nodes, int cap);
/* Graph mode: per-dispatch timing INSIDE a replayed HIP graph.  (External event-record nodes are refused under stream capture by
 * this ROCm build, so:) while a stamp buffer is set, a one-thread kernel in front of and behind every tagged launch writes the
 * 100 MHz wall clock into the caller's device buffer `buf` (cap 64-bit words; 2 per tagged launch + 2) - plain kernel nodes,
 * captured with the step and re-run by every replay.  Record 0 is a calibration pair with nothing in between (one dependent-
 * launch gap + the stamp kernel's run time); singa_prof_read_stamps subtracts it.  NULL switches the mode off. */
int singa_prof_stamps(unsigned long long* buf, int cap);
int singa_prof_read_stamps(const unsigned long long* host_stamps, float* ms, int* tags, int* edges, int* nodes, int cap);
int singa_prof_reset(void);
int singa_calib_copy(const float* src, float* dst, long long n, void* stream);
/* a copy with 16 bytes per lane (n % 4 == 0, 16-byte aligned), `blocks` workgroups of 256 threads moving contiguous chunks of
 * 256 * unroll float4s (unroll in {1, 2, 4, 8} loads in flight per lane): bench.py sweeps a few shapes and reports the best as the
 * practical HBM ceiling of the box */
int singa_calib_copy16(const float* src, float* dst, long long n, int blocks, int unroll, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SINGA_HIP_H */
