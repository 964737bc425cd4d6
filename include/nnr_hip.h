/* libnnr_hip.so -- C-ABI of the MI355X-native NNR training hot path (gfx950 only).
 *
 * The reference (Veason-silverbullet/NNR) is pure Python; the native work on its hot path happens inside
 * third-party libraries that its modules call: ATen/cuDNN behind nn.LSTM / nn.Embedding / nn.Linear / bmm /
 * softmax, torch_scatter 2.0.9, and NCCL.  Each entry point below replaces
 * one such call site; the citation is the reference file:line whose arithmetic it implements.
 *
 * Conventions: extern "C"; plain pointers and sizes; every buffer is a caller-allocated
 * DEVICE pointer (the host side uses PyTorch's caching allocator purely as a memory manager); row-major,
 * contiguous unless a leading dimension is passed; kernels are enqueued on the caller's stream and never
 * synchronise with the host (token counts that depend on the data stay in device memory); returns NNR_OK or a
 * negative error code, never throws.  fp32 everywhere (the parity bar is 1e-4 fp32 on logits/loss).
 */
#ifndef NNR_HIP_H
#define NNR_HIP_H
#include <stdint.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNR_OK 0
#define NNR_ERR_ARG (-1)
#define NNR_ERR_LAUNCH (-2)
#define NNR_ERR_UNSUPPORTED (-3)

int nnr_version(void);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = epilogue(alpha * A_op[M,K] . B_op[K,N]).  Replaces every nn.Linear / torch.bmm on the path
 * (newsEncoders.py:122-123 input projection inside nn.LSTM, :128-129 title_H/title_M; layers.py:168 affine1,
 * :197 K/Q, :286 bmm(graph, x) and W; userEncoders.py:85-91) and their autograd backward GEMMs.
 *   trans_a = 0: A is [M,K], K contiguous (lda).        trans_a = 1: A is [K,M], M contiguous.
 *   trans_b = 0: B is [N,K], K contiguous (nn.Linear weight layout).   trans_b = 1: B is [K,N], N contiguous.
 *   Supported pairs: (0,0) forward, (0,1) data-gradient, (1,1) weight-gradient.
 * Epilogue order per element: x = alpha*acc; += bias[n]; += rowvec[map[m]][n]; act; aux_out = x; *= mul[m][n];
 * += resid[m][n]; dropout (drop_target 3); rowdot += rowdot_w[n]*x; store / accumulate / atomicAdd to C row c_idx[m].
 */
typedef struct nnr_gemm_args {
  const float* A;
  const float* B;
  float* C;                 /* may be NULL when only rowdot_out / aux_out are wanted */
  int M, N, K;
  int lda, ldb, ldc;
  int trans_a, trans_b;
  const int* dyn_dev;       /* device int32: actual extent of the token dimension (<= the static one) */
  int dyn_dim;              /* 0 none, 1: bounds M, 2: bounds K */
  const int* a_idx;         /* trans_a=0: A row m is read from A[a_idx[m]] (negative: zero row); embedding gather */
  const int* b_idx;         /* trans_b=1: B k-row is read from B[b_idx[k]] (negative: zero row) */
  int drop_target;          /* 0 none; 1 gathered-A element (m,k); 2 gathered-B element (k,n); 3 C element (m,n);
                               4 scattered atomic C element (m,n).  Element id = row * drop_cols + col. */
  float drop_p;
  uint32_t drop_seed;
  int drop_cols;
  float alpha;
  const float* bias;        /* [N] */
  const float* rowvec;      /* [*, ldrv] */
  int ldrv;
  const int* rowvec_map;    /* [M] or NULL */
  int act;                  /* 0 none, 1 relu, 2 tanh, 3 sigmoid */
  float* aux_out;           /* [M, ldaux]: value right after the activation */
  int ldaux;
  const float* mul;         /* [M, ldmul] */
  int ldmul;
  const float* resid;       /* [M, ldres] */
  int ldres;
  int accumulate;           /* 1: C += result (after the epilogue); 2: add the old C BEFORE bias/activation (k-split convs) */
  int atomic;               /* atomicAdd into C (implied by split_k > 1) */
  const int* c_idx;         /* [M]: destination row of C for row m (negative: skip) */
  int split_k;              /* > 1: reduction split over blockIdx.z, atomicAdd into a pre-zeroed C */
  const float* rowdot_w;    /* [N]; needs N <= 208 */
  float* rowdot_out;        /* [M] */
  int batch;                /* > 1: blockIdx.z batches with the strides below */
  long strideA, strideB, strideC, stride_aux, stride_res;
  int k_chunk;              /* > 0: like split_k but with fixed-size slices of k_chunk reduction rows (multiple of 32): the number of
                               live slices follows the device-side K; atomicAdd into a pre-zeroed / running C */
  float* colsum_out;        /* trans_a only: colsum_out[m] += sum_k A[k][m]  (fused bias gradient, f32 atomics) */
  int tile;                 /* 0 auto, 1: 256x80, 2: 64x80, 3: 128x208, 4: 128x80, 5: 128x80 with BK=32, 6: 64x80 with BK=64,
                             * 7: 16x80 skinny (K split over the waves; plain NT / NN launches only) */
  /* filled by the library */
  uint32_t drop_thresh;
  float drop_scale;
  int vec_epi;
  int sched;                /* token-reduction (split-K) launches: bits 0-1 how workgroups are dealt to the XCDs, bits 2-7 / 8+ tuning
                             * overrides of the device-side slice count (NNR_TN_DEAL / NNR_TN_WANT / NNR_TN_STAGES) */
  /* reproducible split-K (trans_a = trans_b = 1, split_k > 1, N % 4 == 0): with `slab` set, slice z STORES its partial M x N result to
   * slab[z] and a second launch adds the live slices in slice order into C (and the fused column sums into colsum_out): no f32
   * atomics inside the reduction, bit-identical gradients from run to run (config.py:125-130).  slab_floats >= split_k * (M * N + M). */
  float* slab;
  long slab_floats;
  int slab_mode;            /* filled by the library */
  /* fused cross-selective gate backward (newsEncoders.py:128-131: Ht = H * G, G = sigmoid(pre)) in the epilogue of the GEMM that
   * completes dHt:  x = alpha * acc + pre_add[m][n];  aux_out[m][n] = x * resid * mul * (1 - mul)  (= d pre, resid = H, mul = G);
   * C[m][n] = x * mul  (= dH).  Needs the float4 epilogue (aligned operands, N % 4 == 0), no bias / activation / dropout. */
  const float* pre_add;     /* [M, ldpre]: added to the product before anything else (also without gate_bwd) */
  int ldpre;
  int gate_bwd;
  /* tile = 50 computes the same NT product on the BF16 matrix pipe as six exact bf16 x bf16 products with fp32 accumulation (an fp32 value is
   * exactly the sum of three bf16 values; error vs fp64 a third of the fp32-MFMA kernels'; round 5: measured, round 6: the host's default for
   * NT launches whose B is a weight).  Non-finite operands give non-finite outputs (+-Inf or NaN, not necessarily fp32's choice of the two); finite values up to FLT_MAX are
   * exact (csrc/gemm.hip: split3_bf16, split3_trunc8).  B3: the weight matrix B [N, K] pre-split by nnr_split_bf16x3 into three bf16 images [N, ldb3] (ldb3 % 8 == 0, zero-padded), image i at
   * B3 + i * b3_stride elements. */
  const void* B3;
  long b3_stride;
  int ldb3;
} nnr_gemm_args;

int nnr_gemm_f32(const nnr_gemm_args* args, hipStream_t stream);
/* w [rows, cols] (row stride ld) -> three bf16 images out3[i * img_stride + r * ldo + c] with w == image0 + image1 + image2 EXACTLY (columns
 * cols .. ldo-1 zero).  For nnr_gemm_args.B3 (experimental tile 50); weights change once per optimizer step. */
int nnr_split_bf16x3(const float* w, int rows, int cols, int ld, int ldo, void* out3, long img_stride, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ sequence planner
 * Replaces newsEncoders.py:106-120 (mask[:,0]=1 in place, lengths, torch.sort x2, index_select, pack_padded_sequence and
 * its sorted_length.cpu() host sync).  perm_in (optional, [n]): sorted position -> original row, when the caller wants a
 * specific tie order (e.g. the installed torch's unstable CPU sort); NULL = stable descending order computed on device.
 * Outputs: len_out[n], order[n], rank[n], slen[n], bs[L], off[L+1] (off[L] = #valid tokens, used as dyn_dev by the GEMMs),
 * row_seq[n*L], tok[n*L] (token id per packed row; NULL to skip), prev_f[n*L], prev_r[n*L]. */
int nnr_seq_plan(uint8_t* mask, const int* ids, int n, int L, const int* perm_in, int* len_out, int* order, int* rank, int* slen,
                 int* bs, int* off, int* row_seq, int* tok, int* prev_f, int* prev_r, hipStream_t stream);

/* The candidate call and the history call of one training step (model.py:123-125: the same news encoder applied to
 * [B, 1+neg] and to [B, max_history] news) planned as ONE packed token stream of n0 + n1 sequences, so that every per-token
 * kernel of the step runs once over both calls.  Rows [0, n0) of every per-sequence array come from mask0 / ids0, rows
 * [n0, n0+n1) from mask1 / ids1; both masks get the mask[:,0]=1 fix in place.  Other arguments as nnr_seq_plan. */
int nnr_seq_plan_pair(uint8_t* mask0, const int* ids0, int n0, uint8_t* mask1, const int* ids1, int n1, int L, const int* perm_in,
                      int* len_out, int* order, int* rank, int* slen, int* bs, int* off, int* row_seq, int* tok, int* prev_f,
                      int* prev_r, hipStream_t stream);

/* CNE's rank pairing (newsEncoders.py:112-115,128-129: the title sequence at sorted position r of a call is gated by the cell
 * state of the CONTENT sequence at sorted position r of the same call, and vice versa) for two calls planned as one union
 * stream: order_t / order_c = nnr_seq_plan_pair's `order` of the title / content stream; pm_t[s] = position in the content
 * stream's union order of the partner of the title sequence at position s, pm_c[s] the reverse (pm_c = pm_t^-1).
 * tmp: 4 * n ints of scratch. */
int nnr_cne_pair_map(const int* order_t, const int* order_c, int n0, int n, int* pm_t, int* pm_c, int* tmp, hipStream_t stream);

/* Packed token rows for the MHSA news encoder (round 5).  cover[i][t] = 1 for t <= the last valid position of title i (all ones for a title
 * with no valid position: its softmax over -1e9 scores is uniform, every position counts -- newsEncoders.py:187-200, layers.py:142-143,171);
 * nnr_seq_plan on `cover` packs exactly the rows that can reach the result, and rowmap[i*L + t] = off[t] + rank[i] (t < len[i], else -1)
 * tells nnr_mhsa_fwd_packed / nnr_mhsa_bwd_packed where position t of title i lives. */
int nnr_mask_cover(const uint8_t* mask, int n, int L, uint8_t* cover, hipStream_t stream);
int nnr_seq_rowmap(const int* off, const int* rank, const int* len, int n, int L, int* rowmap, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ Bi-LSTM
 * Replaces nn.LSTM(bidirectional) on a PackedSequence (newsEncoders.py:66-67, 119-127) and its backward.
 * Gate columns are kept in "p-order": p = (unit/16)*64 + (unit%16)*4 + gate, padded to NP = ceil(H/16)*64 per direction. */
int nnr_lstm_dims(int H, int* UB, int* HP, int* NP);
/* w_ihp [2*NP, E], b_p [2*NP] (= b_ih + b_hh), wf [2*UB*4*UB*256], wb [2*UB*(NP/16)*256]; w_ihp_t (optional) [E, 2*NP] = w_ihp^T,
 * the K-contiguous operand of the embedding-row gradient GEMM */
int nnr_lstm_pack_weights(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f, const float* w_ih_r,
                          const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, int H, int E, float* w_ihp, float* b_p,
                          float* wf, float* wb, float* w_ihp_t, hipStream_t stream);
/* dw_ihp [2*NP, E], db_p [2*NP], dw_hhp [2, NP, H] -> gradients in nn.LSTM's parameter layout.  zero_src != 0: the packed buffers are
 * returned all-zero (a persistent workspace the next step's split-K GEMMs accumulate into again: no per-step fill launches). */
int nnr_lstm_unpack_grads(float* dw_ihp, float* db_p, float* dw_hhp, int H, int E, float* dw_ih_f, float* dw_hh_f,
                          float* db_ih_f, float* db_hh_f, float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                          int accumulate /* 0: overwrite, 1: atomic += */, int zero_src, hipStream_t stream);
typedef struct nnr_lstm_problem {
  const int* bs; const int* off; const int* slen; const int* prev_f; const int* prev_r;   /* from nnr_seq_plan */
  int n, L;
  float* gates;       /* [rows, 2*NP]  fwd: in = x.W_ihp^T + b_p, out = activated gates; bwd: in = gates, out = d(pre-activations) */
  float* cell;        /* [rows, 2*HP]  c_t */
  float* hout;        /* [rows, 2*H]   h_t = [fwd | rev] */
  float* cn;          /* [n, 2*H]      final cell states in sorted order */
  const float* wf;    /* forward fragment-layout W_hh */
  const float* wb;    /* backward fragment-layout W_hh */
  const float* dh;    /* bwd: dL/dH [rows, 2*H] */
  const float* dcn;   /* bwd: dL/dc_n [n, 2*H] or NULL */
  unsigned* sync;     /* optional workspace of nnr_lstm_sync_bytes(n) bytes, ZERO-FILLED ONCE BY THE CALLER before its first launch
                       * and reusable by later launches without clearing (exchange words carry a per-launch epoch; one workspace
                       * must not be shared by two launches that can be in flight together): when every problem of a
                       * launch has one and H = 200, each 16-sequence tile runs on a PAIR of CUs with W_hh resident in
                       * registers/LDS, exchanging half of h_t per step; after the launch the 64 bytes at
                       * nnr_lstm_sync_diag_offset(n) hold diagnostics (word 0 = spin-wait time-outs of that launch, must be 0) */
} nnr_lstm_problem;
size_t nnr_lstm_sync_bytes(int n);
size_t nnr_lstm_sync_diag_offset(int n);   /* byte offset of the diagnostics block inside that workspace */
/* Registers a caller-owned, zero-initialised device counter that every exchange time-out of every later launch adds to (NULL
 * unregisters).  A time-out is a data-poisoning event, not a retry: the waiting lane continues with NaN, the step's loss and
 * gradient norm become NaN, and nnr_clip_adam leaves parameters and moments untouched for such a step.  (The reference has no
 * counterpart: cuDNN's nn.LSTM is one kernel, newsEncoders.py:119-127.) */
int nnr_lstm_set_timeout_counter(unsigned* dev_counter);
/* up to 4 problems (title + content streams of the candidate call and of the history call) run in ONE launch: the
 * recurrence is bound by its longest dependent chain, so independent streams are free to share it */
int nnr_lstm_fwd(const nnr_lstm_problem* probs, int nprob, int H, hipStream_t stream);
int nnr_lstm_bwd(const nnr_lstm_problem* probs, int nprob, int H, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ attention pooling
 * softmax(mask(score)) . x : the tails of `Attention` (layers.py:169-175) and `ScaledDotProduct_CandidateAttention`
 * (layers.py:197-203), forward and backward; see pool.hip for the layouts. */
typedef struct nnr_pool_args {
  const float* x; int ldx; int D; int n; int L;
  int packed;                                /* 1: time-major packed rows (off/slen/order), 0: dense [n, L, D] */
  const int* off; const int* slen; const int* order;
  const uint8_t* mask; int mask_div;         /* dense only: mask[(s / mask_div) * L + t] */
  const float* score;                        /* given scores (per row) ... */
  const float* v; int ldv; float scale;      /* ... or score = scale * <x, v[out index]> */
  float* alpha;
  float* out; int ldo; const float* add_in; int ldadd;     /* out = pooled (+ add_in) */
  const float* dout; int lddo; const float* dout2; int lddo2;   /* backward: upstream = dout (+ dout2) */
  float* dx; int lddx; int dx_accumulate;
  float* dscore;
  float* dv; int lddv;
  /* forward only, instead of `score`: score[row] = <th[row, :A], w2>  (the w2 . tanh(.) of layers.py:168 computed in the pool's
   * own pass over the tokens; A <= 256, A % 4 == 0, th rows laid out like x) */
  const float* th; int ldth; int A; const float* w2;
  /* backward only (round 5): the token gradient of a SECOND pool over the same x, folded into this call's ONE write of dx:
   *   dx[row] = alpha[row] * (dout + dout2)  +  alpha_b[row] * dout_b[out index]  +  scale_b * dscore_b[row] * v_b[out index]
   * CNE pools every token stream twice (self attention, layers.py:167-175, and cross attention, layers.py:196-203, newsEncoders.py:
   * 132-137): the cross pool's backward (which must run first: its dv feeds the OTHER stream's self vector) is called with dx = NULL
   * and only writes dscore / dv; the self pool's backward then writes dH~ once instead of write + read-modify-write.  All NULL / 0: off. */
  const float* alpha_b; const float* dout_b; int lddo_b; const float* dscore_b; const float* v_b; int ldv_b; float scale_b;
} nnr_pool_args;
int nnr_attn_pool_fwd(const nnr_pool_args* a, hipStream_t stream);
int nnr_attn_pool_bwd(const nnr_pool_args* a, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ elementwise / reductions */
int nnr_gate_bwd(const float* dHt, const float* H, const float* G, float* dH, float* dpre, const int* rows_dev, int rows, int cols,
                 hipStream_t stream);                                            /* newsEncoders.py:128-131 backward */
int nnr_packed_seq_sum(const float* x, int D, const int* off, const int* slen, int n, float* out, hipStream_t stream);
/* `ws`: NULL (f32 atomics straight into the destination: arrival order, not reproducible), or a workspace of
 * nnr_slot_workspace_floats(N) floats owned by the calling stream: every workgroup stores the column sums of its share of the live
 * rows to its own slot row, and a second tiny launch adds the slot rows in a fixed order into the destination (one f32 atomic per
 * column and call) -- bit-identical from run to run, and no serialisation on the destination's few cache lines. */
int nnr_slot_workspace_floats(int N);
int nnr_tanh_score_bwd(float* th, const float* ds, const float* w2, float* dw2, const int* rows_dev, int rows, int A, float* ws,
                       hipStream_t stream);                                      /* layers.py:168-169 backward */
int nnr_colsum(const float* x, int ld, const int* rows_dev, int rows, int N, float* out_accum, float* ws, hipStream_t stream);
int nnr_rowdot(const float* x, int ld, const float* w, const int* rows_dev, int rows, int N, float* out,
               hipStream_t stream);                                              /* out[row] = <x[row, :N], w>  (layers.py:168) */
int nnr_small_embed_fwd(const float* table, const int* idx, int n, int dim, float* out, int ldo, float p, uint32_t seed,
                        hipStream_t stream);                                     /* newsEncoders.py:51-53 */
int nnr_small_embed_bwd(const int* idx, int n, int dim, const float* dout, int lddo, float* dtable_accum, float p, uint32_t seed,
                        hipStream_t stream);
/* nn.Embedding forward / backward for a dense id tensor (newsEncoders.py:117-118, 163, 193) with the in-place dropout fused:
 * out[row,:] = dropout(table[idx[row],:]) (negative idx: zero row); dtable[idx[row],:] += mask * dout[row,:] (f32 atomics). */
int nnr_embed_gather(const float* table, const int* idx, long n, const int* n_dev /* optional live row count */, int dim, float* out,
                     float p, uint32_t seed, hipStream_t stream);
int nnr_embed_scatter(const float* dout, const int* idx, long n, int dim, float* dtable_accum, float p, uint32_t seed,
                      hipStream_t stream);
/* the same over the first min(n, *n_dev) rows (packed token streams: the live row count stays on the device) */
int nnr_embed_scatter_dyn(const float* dout, const int* idx, long n, const int* n_dev, int dim, float* dtable, float p, uint32_t seed,
                          hipStream_t stream);
/* Reproducible form of the embedding-row gradient (the reference's runs are seeded and deterministic: config.py:125-130; autograd's
 * nn.Embedding backward at newsEncoders.py:117-118).  nnr_token_sort: stable sort of the live packed rows by word id (rows beyond
 * min(cap, *n_dev) and ids outside [0, vocab) get the pad key `vocab` and sort to the end); keys_tmp / rows_tmp / keys_sorted /
 * rows_sorted: `cap` entries each, temp: nnr_token_sort_workspace_bytes(cap, vocab) bytes.  It needs nothing but the planned token
 * stream, so the host side issues it on a side stream under the forward pass.  nnr_embed_scatter_sorted: dtable[w,:] += sum of
 * mask * dout[row,:] over the rows of word w IN LIST ORDER -- every word receives ONE f32 atomic add per element and launch, so
 * with at most two launches per step into a zeroed table gradient (title stream, content stream) the result is bit-reproducible;
 * partial_ws: nnr_embed_scatter_sorted_workspace_floats(cap) floats; dim <= 320. */
size_t nnr_token_sort_workspace_bytes(long cap, int vocab);
int nnr_token_sort(const int* tok, long cap, const int* n_dev, int vocab, unsigned* keys_tmp, int* rows_tmp, unsigned* keys_sorted,
                   int* rows_sorted, void* temp, size_t temp_bytes, hipStream_t stream);
size_t nnr_embed_scatter_sorted_workspace_floats(long cap);
int nnr_embed_scatter_sorted(const float* dout, const unsigned* keys_sorted, const int* rows_sorted, long cap, int vocab, int dim,
                             float* dtable_accum, float p, uint32_t seed, float* partial_ws, hipStream_t stream);
int nnr_transpose2d(const float* in, float* out, long rows, int cols, int accumulate, hipStream_t stream);
/* `count` independent transposes out[c][r] = in[r][c] in ONE launch; the descriptors live in device memory (the host side keeps
 * them for the W^T copies of the weights that the data-gradient GEMMs multiply by: refreshed once per optimizer step). */
typedef struct nnr_transpose_desc { const float* in; float* out; int rows, cols; } nnr_transpose_desc;
int nnr_transpose_batch(const nnr_transpose_desc* descs_dev, int count, hipStream_t stream);
int nnr_add(float* y, const float* x, long n, float alpha, hipStream_t stream);
int nnr_add_atomic(float* y, const float* x, long n, float alpha, hipStream_t stream);   /* y += alpha*x with f32 atomics */
int nnr_add2d(float* y, int ldy, const float* x, int ldx, int rows, int cols, float alpha, int accumulate, hipStream_t stream);
/* y[b, j, :] = x[b, :] for j < N -- the user vector repeated over the candidates (userEncoders.py:172 `.repeat`, :190 `.expand`) -- and its
 * backward dx[b, :] = sum_j dy[b, j, :] (ascending j).  x / dx [B, D], y / dy [B, N, D], contiguous. */
int nnr_expand_rows_fwd(const float* x, float* y, int B, int N, int D, hipStream_t stream);
int nnr_expand_rows_bwd(const float* dy, float* dx, int B, int N, int D, hipStream_t stream);
int nnr_dropout(const float* x, float* y, long n, float p, uint32_t seed, hipStream_t stream);
int nnr_relu_bwd(const float* dy, const float* y, float* dx, long n, hipStream_t stream);
/* SUE's per-user graph aggregate (layers.py:285-292: `graph @ feature` inside GCNLayer.forward, B users x [G, G] x [G, D]) with
 * GCNLayer's epilogue:  y = dropout(relu?(graph_b . z_b + bias) + resid), r_out (optional) = the value after bias / ReLU (what the
 * backward mask needs).  G <= 128.  p / seed: counter-based dropout over the flat [B, G, D] index (same mask as nnr_relu_drop_bwd). */
int nnr_gcn_aggregate_fwd(const float* graph, const float* z, const float* bias, const float* resid, float* r_out, float* y, int B, int G,
                          int D, int relu, float p, uint32_t seed, hipStream_t stream);
/* Its backward:  ds = mask(dy) * (r > 0), dx0 (optional) = mask(dy) (the residual branch), dz_b = graph_b^T . ds_b.  r == NULL: plain
 * dz_b = graph_b^T . dy_b (ds / dx0 untouched). */
int nnr_gcn_aggregate_bwd(const float* graph, const float* dy, const float* r, float* ds, float* dx0, float* dz, int B, int G, int D, float p,
                          uint32_t seed, hipStream_t stream);
int nnr_relu_drop_bwd(const float* dy, const float* r, float* ds, float* dx, long n, float p, uint32_t seed, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ multi-head self-attention core
 * MultiHeadAttention.forward after the W_Q/W_K/W_V projections (layers.py:137-147) on v_mfma_f32_32x32x2_f32:
 * qkv [n*Lq, 3*heads*dh] = [Q | K | V] (head h at columns h*dh), key mask [n, Lq] (0 -> -1e9) or NULL, scale = 1/sqrt(dh);
 * out [n*Lq, heads*dh]; prob [n*heads, NB*NB*1024] (NB = 1 for Lq <= 32, 2 for Lq <= 64) saved for backward, or NULL in
 * both calls: backward then recomputes the probabilities from Q, K (the product path does this).
 * drop_p > 0 fuses the dropout that follows the attention (newsEncoders.py:196): out = nnr_dropout(O, drop_p, seed) bit for bit
 * (mask = f(seed, flat index in out)), and backward applies the same mask to dout while staging it. */
int nnr_mhsa_fwd(const float* qkv, const uint8_t* mask, int n, int Lq, int heads, int dh, float scale, float* out, float* prob,
                 float drop_p, uint32_t seed, hipStream_t stream);
int nnr_mhsa_bwd(const float* qkv, const uint8_t* mask, const float* prob, const float* dout, int n, int Lq, int heads, int dh,
                 float scale, float* dqkv, float drop_p, uint32_t seed, hipStream_t stream);
/* The same over PACKED token rows: qkv / out / dout / dqkv hold only the rows rowmap names (position t of sample i at row rowmap[i*Lq + t],
 * -1 = no such row: reads as zero, is not stored); `mask` is still the dense [n, Lq] key mask.  Needs heads % 4 == 0, dh % 4 == 0, Lq <= 32
 * rows per 4-head group as in the dense cooperative path; dropout indices are (packed row) * heads*dh + column. */
int nnr_mhsa_fwd_packed(const float* qkv, const uint8_t* mask, const int* rowmap, int n, int Lq, int heads, int dh, float scale, float* out,
                        float drop_p, uint32_t seed, hipStream_t stream);
int nnr_mhsa_bwd_packed(const float* qkv, const uint8_t* mask, const int* rowmap, const float* dout, int n, int Lq, int heads, int dh, float scale,
                        float* dqkv, float drop_p, uint32_t seed, hipStream_t stream);
/* Round 6: PAIRED short titles.  The attention core multiplies 32 x 32 blocks whatever a title's length, and 85 % of MIND-shaped titles cover <= 16
 * positions; nnr_mhsa_pair_map lays the titles of a packed call (Lq = 32) out as virtual samples in the plan's sorted order -- the titles covering
 * more than 16 positions alone, the others two to a sample (positions 0..15 / 16..31) -- and nnr_mhsa_fwd_paired / _bwd_paired run the core over
 * them, with -inf scores between the two titles of a pair (exact zeros in every product that follows).  Same results as nnr_mhsa_*_packed up to
 * the order of fp32 additions inside a softmax row; ~0.6x of its matrix work.  vrowmap / vmask: [n, 32]; off: the plan's offsets (off[17] - off[16]
 * = the number of unpaired titles).  layers.py:132-148. */
int nnr_mhsa_pair_map(const int* off, const int* slen, const int* order, const uint8_t* mask, int n, int L, int* vrowmap, uint8_t* vmask,
                      hipStream_t stream);
int nnr_mhsa_fwd_paired(const float* qkv, const uint8_t* vmask, const int* vrowmap, const int* off, int n, int heads, int dh, float scale,
                        float* out, float drop_p, uint32_t seed, hipStream_t stream);
int nnr_mhsa_bwd_paired(const float* qkv, const uint8_t* vmask, const int* vrowmap, const int* off, const float* dout, int n, int heads, int dh,
                        float scale, float* dqkv, float drop_p, uint32_t seed, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ SUE (userEncoders.py:68-98) */
/* cmask_fix (optional): the [B, Kc + 1] cluster mask; its last column is set to 1 in place by the same launch (:73).
 * dx0_add (optional): second addend of the upstream gradient (the outer residual of :81), dX0 = dx0 + dx0_add. */
int nnr_sue_x0_fwd(const float* hist, const float* proxy, float* x0, int B, int Hn, int Kc, int D, float p, uint32_t seed,
                   uint8_t* cmask_fix, hipStream_t stream);                      /* :80 */
int nnr_sue_x0_bwd(const float* dx0, const float* dx0_add, float* dhist, float* dproxy_accum, int B, int Hn, int Kc, int D, float p,
                   uint32_t seed, hipStream_t stream);
int nnr_sue_slice_fwd(const float* gcn, const float* x0, float* gfeat, int B, int Hn, int G, int D, hipStream_t stream);   /* :81-82 */
int nnr_sue_slice_bwd(const float* dgfeat, float* dpad, int B, int Hn, int G, int D, hipStream_t stream);
/* torch_scatter.scatter_softmax + scatter_sum (userEncoders.py:85-89): kf [B,Hn,A], qc [B,N,A], g [B,Hn,D], cidx int64 [B,Hn]
 * -> alpha [B,N,Hn], feat [B,N,C,D] */
int nnr_sue_intra_fwd(const float* kf, const float* qc, const float* g, const long* cidx, int B, int N, int Hn, int C, int A, int D,
                      float* alpha, float* feat, hipStream_t stream);
int nnr_sue_intra_bwd(const float* kf, const float* qc, const float* g, const long* cidx, const float* alpha, const float* dfeat, int B,
                      int N, int Hn, int C, int A, int D, float* dg, float* dkf, float* dqc, float* ds_ws /* [B*N*Hn] scratch */,
                      hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ device-resident corpus
 * (SURVEY.md section 8 f-1 / f-2).  The corpus tables MIND_Corpus builds (MIND_corpus.py:261-268, 336-353) live in HBM;
 * a training batch is described by behaviour indices + the (1 + K) sampled news ids of each behaviour.
 * nnr_corpus_batch   = MIND_Train_Dataset.__getitem__ + default collate (MIND_dataset.py:70-76): fills the 21 batch tensors
 *                      (dtypes of the reference's DataLoader: int64 user ids / cluster indices, int32 ids, 1-byte bools, fp32 graph).
 *                      With graph_table == NULL the graph / cluster mask / cluster indices are left to nnr_history_graph.
 * nnr_history_graph  = MIND_Corpus.preprocess step 6 (MIND_corpus.py:162-221) for a batch, from the history's category ids:
 *                      norm 0 = none, 1 = symmetric D^-1/2 A D^-1/2, 2 = asymmetric D^-1 A (self connections on), 3 = none and
 *                      NO self connections (--no_self_connection; the reference asserts it excludes normalisation,
 *                      config.py:56,111).  Bit-identical to the numpy result. */
typedef struct nnr_corpus_tables {
  const int* news_category; const int* news_subCategory;                       /* [news] */
  const int* title_text; const uint8_t* title_mask; const int* title_entity;   /* [news, T] */
  const int* abstract_text; const uint8_t* abstract_mask; const int* abstract_entity;   /* [news, C] */
  const long* beh_user; const int* beh_history; const uint8_t* beh_history_mask; const int* beh_line;   /* [behaviours(, H)] */
  const float* graph_table; const uint8_t* cmask_table; const long* cidx_table;   /* optional [lines, G, G] / [lines, K1] / [lines, H] */
  int T, C, H, G, K1;                                                          /* G = H + category_num, K1 = category_num + 1 */
} nnr_corpus_tables;
typedef struct nnr_batch_out {                                                 /* the 21 tensors, argument order of trainer.py:105-106 */
  long* user_id;
  int* u_cat; int* u_sub; int* u_tt; uint8_t* u_tm; int* u_te; int* u_ct; uint8_t* u_cm; int* u_ce;
  uint8_t* u_hmask; float* u_graph; uint8_t* u_cmask; long* u_cidx;
  int* n_cat; int* n_sub; int* n_tt; uint8_t* n_tm; int* n_te; int* n_ct; uint8_t* n_cm; int* n_ce;
} nnr_batch_out;
int nnr_corpus_batch(const nnr_corpus_tables* t, const nnr_batch_out* o, const int* beh_idx, const int* samples, int ld_samples, int B,
                     int S, hipStream_t stream);
int nnr_history_graph(const int* cats, const uint8_t* hmask, int B, int H, int K, int norm, float* graph, uint8_t* cmask, long* cidx,
                      hipStream_t stream);
/* Evaluation tail: util.py:50-59 (per-impression ranks by descending score, ties in file order) +
 * evaluate.py:8-29,76-81 (AUC, MRR, nDCG@5, nDCG@10 per impression, float64).  offsets [n_impressions + 1] delimit the
 * (contiguous) candidates of each impression; per_impression [n_impressions, 4]; NaN row for a single-class impression. */
int nnr_rank_metrics(const float* scores, const uint8_t* labels, const long* offsets, int n_impressions, int* ranks,
                     double* per_impression, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ click predictor, loss, optimiser */
int nnr_logits_loss_fwd(const float* user, const float* cand, int B, int N, int D, float* logits, float* loss, float* dlogits,
                        hipStream_t stream);                                     /* model.py:126-127, trainer.py:64-66 */
int nnr_logits_fwd(const float* user, const float* cand, int B, int N, int D, float* logits, hipStream_t stream);   /* model.py:127 */
int nnr_nls_loss(const float* logits, int B, int N, float* loss, float* dlogits, hipStream_t stream);               /* trainer.py:64-66 */
int nnr_logits_bwd(const float* dlogits, const float* user, const float* cand, int B, int N, int D, float* duser, float* dcand,
                   int dcand_accumulate, hipStream_t stream);
/* --gcn_layer_norm (config.py:61; layers.py:273-274,287-288): nn.LayerNorm([D]) between the graph convolution and the ReLU,
 * fused with the rest of GCNLayer.forward / GCN.forward:  y = dropout(relu(LN(u) * gamma + beta) + resid)  (resid may be NULL,
 * p may be 0).  xhat [rows, D], rstd [rows] and r_out = relu(.) [rows, D] are saved for the backward pass; D <= 1024.
 * Backward: du = d(LN input) from dv = d(LN output); dgamma / dbeta are ACCUMULATED (f32 atomics). */
int nnr_layernorm_fwd(const float* u, const float* gamma, const float* beta, float eps, long rows, int D, float* xhat, float* rstd,
                      float* r_out, const float* resid, float* y, float p, uint32_t seed, hipStream_t stream);
int nnr_layernorm_bwd(const float* dv, const float* xhat, const float* rstd, const float* gamma, long rows, int D, float* du,
                      float* dgamma, float* dbeta, hipStream_t stream);
/* *out = sum g^2 (stored), a DETERMINISTIC function of g (fixed-order two-level sum: data-parallel ranks must clip by the same bits);
 * launches of one process must be stream-ordered (one device-global scratch) */
int nnr_sumsq(const float* g, long n, float* out, hipStream_t stream);
/* The same over one SPAN of the gradient, chained: *out = sum g^2 + (add_in ? *add_in : 0), fixed order.  `slot` (0..3) selects the scratch
 * set: launches that may run concurrently (different streams) must use different slots; nnr_sumsq uses slot 0.  Round 5: the norm of the
 * word-embedding table's gradient (70 % of the floats) is taken on a helper stream as soon as its last scatter is done, beside the LSTM
 * weight-gradient GEMMs of the step's tail, and only the remaining spans are summed on the optimizer's stream (trainer.py:118). */
int nnr_sumsq_part(const float* g, long n, float* out, const float* add_in, int slot, hipStream_t stream);
/* clip_grad_norm_(max_norm = clip) + torch.optim.Adam step on one flat buffer (trainer.py:118-120); grads are scaled by
 * grad_scale first (1/world_size after the RCCL sum all-reduce).  A step whose squared gradient norm is not finite is
 * skipped as a whole (parameters and moments untouched). */
int nnr_clip_adam(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float grad_scale, float clip, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, hipStream_t stream);

/* Steps nnr_clip_adam skipped so far in this process (non-finite gradient norm: overflow, or the NaN poison of a timed-out recurrence
 * exchange).  The reference would go visibly NaN there (trainer.py:118-120); here the step is dropped and COUNTED -- poll this every few
 * hundred steps (synchronous device read); reset != 0 clears the counter. */
int nnr_adam_skipped_steps(unsigned* host_out, int reset);
/* The same count WITHOUT a synchronisation: nnr_clip_adam's kernel mirrors it into pinned, device-mapped host memory, so the value covers
 * every optimizer step that has completed on the device.  Cheap enough to read after every step (Trainer.train_step does; with an event
 * wait every NNR_SKIP_POLL-th step that bounds how far the host runs ahead, a skipped step is reported within 2 x NNR_SKIP_POLL steps). */
int nnr_adam_skipped_peek(unsigned* host_out);

/* ------------------------------------------------------------------------------------------------ data parallelism (RCCL over xGMI)
 * Replaces DistributedDataParallel's gradient all-reduce / parameter broadcast (trainer.py:212-219,297): one communicator per
 * process = per GPU, ONE in-place fp32 sum all-reduce of the flat gradient buffer per step; the 1/world average is folded into
 * nnr_clip_adam.  RCCL is resolved at run time from the copy already loaded in the process (PyTorch-ROCm's).  Rank 0 creates
 * the 128-byte id and the launcher distributes it (any side channel: the torchrun store, a file, MPI). */
typedef struct nnr_dp_ctx nnr_dp_ctx;
int nnr_dp_unique_id(void* out128);
int nnr_dp_init(const void* uid128, int rank, int world, nnr_dp_ctx** ctx);          /* binds to the current HIP device */
int nnr_dp_allreduce(nnr_dp_ctx* ctx, float* flat, size_t n, hipStream_t stream);
int nnr_dp_broadcast(nnr_dp_ctx* ctx, float* flat, size_t n, int root, hipStream_t stream);
int nnr_dp_destroy(nnr_dp_ctx* ctx);
/* Test hook for boxes with ONE GPU (RCCL refuses two ranks on one device): from now on every nnr_dp_allreduce of `ctx` behaves as if
 * `ranks` ranks held the SAME buffer -- the sum is ranks x the local values (a scale kernel on the same stream right behind the
 * collective, so it is part of whatever launch sequence the all-reduce is part of, recorded tapes included).  With a power of two the
 * result is exact, and a bucket that was NOT exchanged shows as 1 x instead of ranks x its gradient (tests/test_hip_dp_gpu.py:
 * the C-ABI binding + touched-row exchange over replayed steps).  ranks = 1 turns it off.  Never set by the product path. */
int nnr_dp_emulate_ranks(nnr_dp_ctx* ctx, int ranks);
/* Diagnostics: a stand-in for RCCL's RESIDENT ring kernels on a box with one GPU -- `workgroups` x 512 threads sweep their slice of buf
 * [n] `iters` times (read-modify-write through HBM), occupying as many CU slots for the duration.  The co-residency soak of the CU-pair
 * recurrence (tests/test_hip_dp_gpu.py, tools/replay_soak.py --busy) runs it on a side stream beside every step. */
int nnr_dp_busy(float* buf, long n, int workgroups, int iters, hipStream_t stream);
/* Touched-row exchange of the word-embedding table's gradient (the table is 70 % of the all-reduced bytes, and only
 * the rows of words in the step's batch are non-zero; trainer.py:297 reduces all of it).  nnr_rows_touch: flags[tok[i]] = 1 for the live
 * packed rows of a token stream (flags: V floats, zeroed by the caller at the start of the step and summed over the ranks before
 * nnr_rows_compact); nnr_rows_compact: pos[w] = index of row w among the touched rows (flags[w] > 0) or -1, *count = their number;
 * nnr_rows_pack / nnr_rows_unpack: packed[pos[w], :] <-> dense[w, :] for the touched rows.  The dense gradient after unpacking equals
 * what the full all-reduce produces (untouched rows are zero on every rank), so clip + Adam stay dense and unchanged. */
int nnr_rows_touch(const int* tok, long cap, const int* n_dev, int V, float* flags, hipStream_t stream);
int nnr_rows_compact(const float* flags, int V, int* pos, int* count, hipStream_t stream);
int nnr_rows_pack(const float* dense, const int* pos, int V, int E, float* packed, hipStream_t stream);
int nnr_rows_unpack(const float* packed, const int* pos, int V, int E, float* dense, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ fused small launches (csrc/fuse.hip)
 * feature_fusion (newsEncoders.py:50-54) for the union of the candidate call (rows [0, n0), ids cat0 / sub0) and the history call
 * (rows [n0, n0 + n1), ids cat1 / sub1; n1 may be 0): out[row, 0:cd] = dropout(category row), out[row, cd:cd+sd] = dropout(subCategory
 * row); masks = nnr_small_embed_fwd's (flat index row * dim + column, one seed per table).  Backward accumulates both table gradients. */
int nnr_fusion_rows_fwd(const float* cat_table, const float* sub_table, const int* cat0, const int* sub0, int n0, const int* cat1,
                        const int* sub1, int n1, int cd, int sd, float* out, int ldo, float p, uint32_t seed_cat, uint32_t seed_sub,
                        hipStream_t stream);
int nnr_fusion_rows_bwd(const int* cat0, const int* sub0, int n0, const int* cat1, const int* sub1, int n1, int cd, int sd, const float* dout,
                        int lddo, float* dcat_table_accum, float* dsub_table_accum, float p, uint32_t seed_cat, uint32_t seed_sub,
                        hipStream_t stream);
/* The same gradients, reproducibly (one workgroup per table row scans the ids in order: no run merging, no arrival-order atomics);
 * ncat / nsub = rows of the two tables; cd, sd <= 128. */
int nnr_fusion_rows_bwd_det(const int* cat0, const int* sub0, int n0, const int* cat1, const int* sub1, int n1, int cd, int sd, int ncat,
                            int nsub, const float* dout, int lddo, float* dcat_table_accum, float* dsub_table_accum, float p,
                            uint32_t seed_cat, uint32_t seed_sub, hipStream_t stream);
/* Click predictor + loss + their backward in one launch (model.py:126-127, trainer.py:64-66): logits [B, N], loss (scalar, a fixed-order
 * mean), dlogits [B, N] (optional), duser / dcand [B, N, D] (both or neither); terms_ws: B floats of scratch.  N <= 64. */
int nnr_click_loss(const float* user, const float* cand, int B, int N, int D, float* logits, float* loss, float* dlogits, float* duser,
                   float* dcand, float* terms_ws, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------ fills / copies
 * What the host framework's fill / copy / index-put kernels did inside the step (optimizer.zero_grad() at trainer.py:116,
 * torch.cat of the two encoder calls' id tensors, `user_history_category_mask[:, -1] = 1` at userEncoders.py:73), as entry points,
 * so that a whole training step is a sequence of calls into this library (see the tape below). */
int nnr_fill_zero(void* p, size_t bytes, hipStream_t stream);
int nnr_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream);          /* device to device */
int nnr_fill_column_u8(uint8_t* m, int rows, int cols, int col, int value, hipStream_t stream);   /* m[:, col] = value */

/* ------------------------------------------------------------------------------------------------ launch-sequence tape
 * Replaces the per-call Python dispatch of the reference's training step (trainer.py:105-120: ~140 framework calls per step)
 * by a native replay: the host side records the entry-point calls of ONE step -- function, arguments (by-pointer structs are
 * copied), HIP stream -- plus the step's cross-stream dependencies, and replays the sequence with one call per segment
 * (csrc/tape.hip).  Every replayed call runs the real entry point above.  Per-step changes are patched in before a replay:
 * value patches (dropout seeds, Adam's step number: value[kind] + addend) and input patches (pointers into the batch tensors:
 * input[kind - 1000] + addend).  A tape owns nothing but its argument copies and events: buffers stay caller-owned and must
 * outlive it.  Not thread-safe; one tape is replayed by one host thread. */
typedef struct nnr_tape nnr_tape;
int nnr_tape_create(nnr_tape** out);
int nnr_tape_destroy(nnr_tape* t);
int nnr_tape_fn_id(const char* entry_point_name);      /* >= 0, or -1: not recordable (no stream argument / host-only query) */
int nnr_tape_fn_nargs(int fn);                         /* arguments before the trailing hipStream_t */
int nnr_tape_call(nnr_tape* t, int fn, hipStream_t stream, const uint64_t* slots, int nslots, const int* blob_slot,
                  const void* const* blob_ptr, const size_t* blob_bytes, int nblobs, int tag, size_t* slot_off_out, size_t* blob_off_out);
int nnr_tape_wait_stream(nnr_tape* t, hipStream_t waiter, hipStream_t signaller);
int nnr_tape_event_record(nnr_tape* t, uint64_t key, hipStream_t s);
int nnr_tape_event_wait(nnr_tape* t, hipStream_t s, uint64_t key);
int nnr_tape_segment(nnr_tape* t);
int nnr_tape_patch(nnr_tape* t, size_t arena_byte_off, int kind, int width, int64_t addend);
int nnr_tape_finalize(nnr_tape* t);
/* Create the HIP events of timing sets [0, nsets) ahead of a timed window of replays (host-side only, no launch). */
int nnr_tape_prepare_timing(nnr_tape* t, int nsets);
int nnr_tape_info(const nnr_tape* t, int* calls, int* ops, int* segments, int* streams, size_t* arena_bytes);
int nnr_tape_replay(nnr_tape* t, int segment, const uint64_t* values, int nvalues, const uint64_t* inputs, int ninputs, int timing_set);
int nnr_tape_timings(nnr_tape* t, int set, float* ms, int n);
int nnr_tape_timeline(nnr_tape* t, int set, float* start_ms, float* dur_ms, int* stream_idx, int n);
int nnr_tape_last_error(const nnr_tape* t, int* rc, int* call, char* name, int name_cap);

#ifdef __cplusplus
}
#endif
#endif
