// Data side of the hot path, device resident.
//
// * corpus_batch_kernel : MIND_Train_Dataset.__getitem__ + default collate (MIND_dataset.py:70-76) as ONE gather kernel over
//   corpus tables that live in HBM: a batch is described by behaviour indices + (1 + K) sampled news ids per behaviour
//   (a few hundred bytes over PCIe instead of 98.6 KB per impression), one wave per news slot, one per impression for the
//   per-behaviour fields.  HBM-bound byte work: 1.45 KB read + 1.45 KB written per news slot (T = 32, C = 128).
// * history_graph_kernel : the per-behaviour user-history graph, cluster mask and cluster indices that the reference
//   pre-computes into a [behaviours, G, G] fp32 array (MIND_corpus.py:162-221; 18.5 KB per behaviour line, tens of GB of
//   host memory on MIND-large) built on the fly from the history's category ids: one workgroup per impression, the
//   G x G adjacency assembled in LDS, normalised with correctly rounded fp32 div / sqrt / mul (no contraction) so the result is
//   bit-identical to numpy's, written once (18.5 KB).
#include "common.h"

namespace {

struct CorpusTables {
  const int* news_category; const int* news_subCategory;
  const int* title_text; const uint8_t* title_mask; const int* title_entity;
  const int* abstract_text; const uint8_t* abstract_mask; const int* abstract_entity;
  const long* beh_user; const int* beh_history; const uint8_t* beh_history_mask; const int* beh_line;
  const float* graph_table; const uint8_t* cmask_table; const long* cidx_table;    // optional pre-built graphs (else null)
  int T, C, H, G, K1;
};
struct BatchOut {
  long* user_id;
  int* u_cat; int* u_sub; int* u_tt; uint8_t* u_tm; int* u_te; int* u_ct; uint8_t* u_cm; int* u_ce;
  uint8_t* u_hmask; float* u_graph; uint8_t* u_cmask; long* u_cidx;
  int* n_cat; int* n_sub; int* n_tt; uint8_t* n_tm; int* n_te; int* n_ct; uint8_t* n_cm; int* n_ce;
};

template <typename T>
__device__ __forceinline__ void copy_row(T* dst, const T* src, int n, int lane) {
  for (int i = lane; i < n; i += 64) dst[i] = src[i];
}

// grid = B * (H + S) news slots + B per-impression slots; block = one wave
__global__ __launch_bounds__(64) void corpus_batch_kernel(CorpusTables t, BatchOut o, const int* __restrict__ beh_idx,
                                                          const int* __restrict__ samples, int ld_samples, int B, int S) {
  const int lane = threadIdx.x;
  const int slots = B * (t.H + S);
  const int bid = blockIdx.x;
  if (bid < slots) {
    const int b = bid / (t.H + S), k = bid - b * (t.H + S);
    const int beh = beh_idx[b];
    const bool hist = k < t.H;
    const int news = hist ? t.beh_history[(long)beh * t.H + k] : samples[(long)beh * ld_samples + (k - t.H)];
    const long dst = hist ? (long)b * t.H + k : (long)b * S + (k - t.H);
    int* cat = hist ? o.u_cat : o.n_cat; int* sub = hist ? o.u_sub : o.n_sub;
    int* tt = hist ? o.u_tt : o.n_tt; uint8_t* tm = hist ? o.u_tm : o.n_tm; int* te = hist ? o.u_te : o.n_te;
    int* ct = hist ? o.u_ct : o.n_ct; uint8_t* cm = hist ? o.u_cm : o.n_cm; int* ce = hist ? o.u_ce : o.n_ce;
    if (lane == 0) { cat[dst] = t.news_category[news]; sub[dst] = t.news_subCategory[news]; }
    copy_row(tt + dst * t.T, t.title_text + (long)news * t.T, t.T, lane);
    copy_row(te + dst * t.T, t.title_entity + (long)news * t.T, t.T, lane);
    copy_row(ct + dst * t.C, t.abstract_text + (long)news * t.C, t.C, lane);
    copy_row(ce + dst * t.C, t.abstract_entity + (long)news * t.C, t.C, lane);
    if (((t.T | t.C) & 3) == 0) {      // mask rows are whole dwords
      copy_row(reinterpret_cast<uint32_t*>(tm + dst * t.T), reinterpret_cast<const uint32_t*>(t.title_mask + (long)news * t.T), t.T >> 2, lane);
      copy_row(reinterpret_cast<uint32_t*>(cm + dst * t.C), reinterpret_cast<const uint32_t*>(t.abstract_mask + (long)news * t.C), t.C >> 2, lane);
    } else {
      copy_row(tm + dst * t.T, t.title_mask + (long)news * t.T, t.T, lane);
      copy_row(cm + dst * t.C, t.abstract_mask + (long)news * t.C, t.C, lane);
    }
    return;
  }
  const int b = bid - slots;
  if (b >= B) return;
  const int beh = beh_idx[b];
  if (lane == 0) o.user_id[b] = t.beh_user[beh];
  copy_row(o.u_hmask + (long)b * t.H, t.beh_history_mask + (long)beh * t.H, t.H, lane);
  if (t.graph_table) {               // pre-built graphs resident in HBM (the reference's layout): plain row gather
    const int line = t.beh_line[beh];
    copy_row(o.u_graph + (long)b * t.G * t.G, t.graph_table + (long)line * t.G * t.G, t.G * t.G, lane);
    copy_row(o.u_cmask + (long)b * t.K1, t.cmask_table + (long)line * t.K1, t.K1, lane);
    copy_row(o.u_cidx + (long)b * t.H, t.cidx_table + (long)line * t.H, t.H, lane);
  }
}

constexpr int GMAX = 96;
// cats [B, H] : category of every history slot (the batch's user_category); hmask [B, H] : slot is real history
__global__ __launch_bounds__(256) void history_graph_kernel(const int* __restrict__ cats, const uint8_t* __restrict__ hmask, int B, int H,
                                                            int K, int norm, float* __restrict__ graph, uint8_t* __restrict__ cmask,
                                                            long* __restrict__ cidx) {
  __shared__ float A[GMAX * GMAX];
  __shared__ float dsc[GMAX];
  __shared__ int cat[GMAX];
  __shared__ int nh;
  const int b = blockIdx.x, tid = threadIdx.x, G = H + K;
  if (tid == 0) nh = 0;
  __syncthreads();
  for (int i = tid; i < H; i += 256) {
    cat[i] = cats[(long)b * H + i];
    if (hmask[(long)b * H + i]) atomicAdd(&nh, 1);        // the mask is a prefix (MIND_corpus.py:352-353): count = history length
  }
  const float diag = norm == 3 ? 0.f : 1.f;                                                  // :180-183 --no_self_connection: zero diagonal
  for (int i = tid; i < G * G; i += 256) A[i] = ((i / G) == (i % G)) ? diag : 0.f;           // :183 identity (self connections)
  __syncthreads();
  const int n = nh;
  for (int i = tid; i < K + 1; i += 256) cmask[(long)b * (K + 1) + i] = 0;
  for (int i = tid; i < H; i += 256) cidx[(long)b * H + i] = (i < n) ? (long)cat[i] : (long)K;   // :185,:192
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int c = cat[i];
    cmask[(long)b * (K + 1) + c] = 1;                                                        // :191 (same value from every writer)
    A[i * G + H + c] = 1.f;                                                                  // :194-195
    A[(H + c) * G + i] = 1.f;
  }
  for (int p = tid; p < n * n; p += 256) {
    const int i = p / n, j = p - i * n;
    if (j <= i) continue;
    const int ci = cat[i], cj = cat[j];
    if (ci == cj) { A[i * G + j] = 1.f; A[j * G + i] = 1.f; }                                // :199-200
    else { A[(H + ci) * G + H + cj] = 1.f; A[(H + cj) * G + H + ci] = 1.f; }                // :202-203
  }
  __syncthreads();
  float* out = graph + (long)b * G * G;
  if (n == 0 || norm == 0 || norm == 3) {                                                                 // :186 empty history stays un-normalised
    for (int i = tid; i < G * G; i += 256) out[i] = A[i];
    return;
  }
  for (int i = tid; i < G; i += 256) {
    float sm = 0.f;
    for (int j = 0; j < G; ++j) sm += A[i * G + j];                                          // small integers: exact
    // plain `/` and sqrtf are correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt (the
    // __fdiv_rn / __fsqrt_rn intrinsics map to the 1-ulp native instructions and differed from numpy in 0.9 % of entries)
    const float inv = 1.f / sm;
    dsc[i] = (norm == 2) ? inv : sqrtf(inv);                                                 // :207 / :212
  }
  __syncthreads();
  for (int p = tid; p < G * G; p += 256) {
    const int i = p / G, j = p - i * G;
    const float v = __fmul_rn(dsc[i], A[p]);
    out[p] = (norm == 2) ? v : __fmul_rn(v, dsc[j]);                                         // :209 / :214
  }
}

}  // namespace

extern "C" int nnr_corpus_batch(const nnr_corpus_tables* t, const nnr_batch_out* o, const int* beh_idx, const int* samples, int ld_samples,
                                int B, int S, hipStream_t stream) {
  if (!t || !o || !beh_idx || !samples || B <= 0 || S <= 0) return NNR_ERR_ARG;
  CorpusTables ct;
  ct.news_category = t->news_category; ct.news_subCategory = t->news_subCategory; ct.title_text = t->title_text;
  ct.title_mask = t->title_mask; ct.title_entity = t->title_entity; ct.abstract_text = t->abstract_text;
  ct.abstract_mask = t->abstract_mask; ct.abstract_entity = t->abstract_entity; ct.beh_user = t->beh_user;
  ct.beh_history = t->beh_history; ct.beh_history_mask = t->beh_history_mask; ct.beh_line = t->beh_line;
  ct.graph_table = t->graph_table; ct.cmask_table = t->cmask_table; ct.cidx_table = t->cidx_table;
  ct.T = t->T; ct.C = t->C; ct.H = t->H; ct.G = t->G; ct.K1 = t->K1;
  if (ct.graph_table && (!ct.cmask_table || !ct.cidx_table || !ct.beh_line)) return NNR_ERR_ARG;
  BatchOut bo;
  bo.user_id = o->user_id; bo.u_cat = o->u_cat; bo.u_sub = o->u_sub; bo.u_tt = o->u_tt; bo.u_tm = o->u_tm; bo.u_te = o->u_te;
  bo.u_ct = o->u_ct; bo.u_cm = o->u_cm; bo.u_ce = o->u_ce; bo.u_hmask = o->u_hmask; bo.u_graph = o->u_graph; bo.u_cmask = o->u_cmask;
  bo.u_cidx = o->u_cidx; bo.n_cat = o->n_cat; bo.n_sub = o->n_sub; bo.n_tt = o->n_tt; bo.n_tm = o->n_tm; bo.n_te = o->n_te;
  bo.n_ct = o->n_ct; bo.n_cm = o->n_cm; bo.n_ce = o->n_ce;
  hipLaunchKernelGGL(corpus_batch_kernel, dim3(B * (ct.H + S) + B), dim3(64), 0, stream, ct, bo, beh_idx, samples, ld_samples, B, S);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_history_graph(const int* cats, const uint8_t* hmask, int B, int H, int K, int norm, float* graph, uint8_t* cmask,
                                 long* cidx, hipStream_t stream) {
  if (!cats || !hmask || !graph || !cmask || !cidx || B <= 0) return NNR_ERR_ARG;
  if (H + K > GMAX || norm < 0 || norm > 3) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(history_graph_kernel, dim3(B), dim3(256), 0, stream, cats, hmask, B, H, K, norm, graph, cmask, cidx);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

// ------------------------------------------------------------------------------------------------ ranking metrics (f-4)
// The tail of util.compute_scores (util.py:50-59) and evaluate.scoring (evaluate.py:32-89) on the device: one workgroup per
// impression ranks its candidates by descending score (equal scores keep file order, like Python's stable sort) and
// evaluates AUC (roc_auc_score on 1/rank = correctly ordered (positive, negative) pairs / all pairs), MRR, nDCG@5, nDCG@10
// in float64.  An impression with no click or only clicks has no AUC (sklearn raises): its four values are NaN.
namespace {
__device__ __forceinline__ double block_sum(double v, double* sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void rank_metrics_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ labels,
                                                          const long* __restrict__ offsets, int* __restrict__ ranks,
                                                          double* __restrict__ per_imp) {
  __shared__ double sh[256];
  const int imp = blockIdx.x, tid = threadIdx.x;
  const long o = offsets[imp];
  const int n = (int)(offsets[imp + 1] - o);
  const float* s = scores + o;
  const uint8_t* y = labels + o;
  double pos = 0, auc_num = 0, rr = 0, dcg5 = 0, dcg10 = 0;
  for (int i = tid; i < n; i += 256) {
    const float si = s[i];
    int better = 0, neg_below = 0;
    for (int j = 0; j < n; ++j) {
      const float sj = s[j];
      const bool before = (sj > si) || (sj == si && j < i);        // j is ranked ahead of i
      better += before;
      neg_below += (!before && j != i && !y[j]);                   // negatives ranked after i
    }
    const int rank = better + 1;
    ranks[o + i] = rank;
    if (y[i]) {
      pos += 1;
      auc_num += neg_below;
      rr += 1.0 / rank;
      const double g = 1.0 / log2((double)rank + 1.0);
      if (rank <= 5) dcg5 += g;
      if (rank <= 10) dcg10 += g;
    }
  }
  const double P = block_sum(pos, sh);
  const double A = block_sum(auc_num, sh), R = block_sum(rr, sh), D5 = block_sum(dcg5, sh), D10 = block_sum(dcg10, sh);
  if (tid == 0) {
    const double N = n - P;
    double best5 = 0, best10 = 0;
    for (int k = 0; k < 10 && k < (int)P; ++k) {
      const double g = 1.0 / log2((double)k + 2.0);
      if (k < 5) best5 += g;
      best10 += g;
    }
    const bool ok = P > 0 && N > 0;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    per_imp[4 * imp + 0] = ok ? A / (P * N) : nan;
    per_imp[4 * imp + 1] = ok ? R / P : nan;
    per_imp[4 * imp + 2] = ok ? D5 / best5 : nan;
    per_imp[4 * imp + 3] = ok ? D10 / best10 : nan;
  }
}
}  // namespace

extern "C" int nnr_rank_metrics(const float* scores, const uint8_t* labels, const long* offsets, int n_impressions, int* ranks,
                                double* per_impression, hipStream_t stream) {
  if (!scores || !labels || !offsets || !ranks || !per_impression || n_impressions <= 0) return NNR_ERR_ARG;
  hipLaunchKernelGGL(rank_metrics_kernel, dim3(n_impressions), dim3(256), 0, stream, scores, labels, offsets, ranks, per_impression);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
