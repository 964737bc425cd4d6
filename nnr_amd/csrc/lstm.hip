// Bi-LSTM recurrence for CNE (replaces nn.LSTM on a PackedSequence, newsEncoders.py:119-127, and its autograd
// backward) on gfx950.
//
// Decomposition.  The input projection X.W_ih^T + b for ALL valid tokens of both directions is one big GEMM
// (gemm.hip, with the embedding-row gather fused into its A loader); what remains is the strictly sequential part
//     z_t = xw_t + h_{t-1}.W_hh^T ;  c_t = s(f)c_{t-1} + s(i)tanh(g) ;  h_t = s(o)tanh(c_t).
// One workgroup owns a tile of 16 length-sorted sequences of one direction for ALL their time steps: no
// inter-workgroup synchronisation exists anywhere.  Sorted tiles have near-equal lengths, so a workgroup runs exactly
// max(len) steps -- padded positions are never computed (PackedSequence semantics for free).  Tile 0 holds the longest
// sequences and is dispatched first (LPT order).
//
// Per step the workgroup needs h_{t-1}[16, H] . W_hh^T[H, 4H] on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
// W_hh (640 KB / direction at H=200) cannot live in one CU's LDS, so it is streamed from L2 every step in a
// pre-swizzled FRAGMENT layout (each wave-load is one contiguous 1 KiB of exactly the B operands it needs), while the
// tiny A operand h_{t-1} sits in LDS (XOR-swizzled: one conflict-free ds_read_b128 feeds 4 MFMAs).
// Gate columns are re-ordered [unit-block][unit][gate] ("p-order": p = (unit/16)*64 + (unit%16)*4 + gate) so a lane finds
// i,f,g,o of the SAME (sequence, unit) in its own four accumulators AND as one contiguous float4 in memory: the cell
// update is lane-local (no shuffles, no LDS round trip) and every gate access is a 16-byte load/store, 256 B per 16 lanes.
// Streaming buffers (gates / cell / h) are accessed non-temporally so the four W_hh fragment arrays stay L2-resident, and
// waves raise their priority with the tile's length: the longest tile is the critical path of the whole launch.
// The xw buffer is overwritten in place with the activated gates (saved for backward).
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int swz16(int r16) { return (4 - (r16 >> 2)) & 3; }
// LDS offset (floats) of element (row r, column u) in a K-contiguous [16][ld] tile, ld % 16 == 0
__device__ __forceinline__ int lds_off(int r, int u, int ld) {
  return r * ld + (u & ~15) + 4 * (((u >> 2) & 3) ^ swz16(r)) + (u & 3);
}

// Longer tiles = longer dependent chains: give their waves the matrix pipe first (priority beats age in the arbiter).
__device__ __forceinline__ void set_prio_by_length(int tmax) {
  if (tmax >= 96) __builtin_amdgcn_s_setprio(3);
  else if (tmax >= 64) __builtin_amdgcn_s_setprio(2);
  else if (tmax >= 32) __builtin_amdgcn_s_setprio(1);
}

struct LstmProblem {
  // plan
  const int* bs; const int* off; const int* slen; const int* prev_f; const int* prev_r;
  int n, L;
  // buffers
  float* gates;        // [rows, 2*NP]  in: xw (pre-activation, p-order)   out: activated gates   (bwd: in gates, out dgates)
  float* cell;         // [rows, 2*HP]
  float* hout;         // [rows, 2*H]
  float* cn;           // [n, 2*H]   final cell states, sorted order
  const float* wfrag;  // fwd: Wf [2][UB][4][KG][64][4] ; bwd: Wb [2][UB][NP/16][64][4]
  const float* dh;     // bwd: upstream dL/dH [rows, 2*H]
  const float* dcn;    // bwd: upstream dL/dc_n [n, 2*H] (sorted order) or null
};
struct LstmArgs { LstmProblem p[4]; int nprob; int H; int dbg; };   // dbg: timing-attribution mask (NNR_LSTM_DBG), 0 in production

// ------------------------------------------------------------------------------------------------ forward
template <int UB>
__global__ __launch_bounds__((UB > 4 ? 16 : 4) * 64) void lstm_fwd_kernel(LstmArgs a) {
  constexpr int NW = UB > 4 ? 16 : 4, NT = NW * 64;      // one 16-unit block per wave when the hidden size is large
  // balanced variant kept for reference: under hipcc's scheduling (no room for the fragment double buffer in 128 VGPRs) it
  // measured 2.21 ms vs 2.08 ms for the one-block-per-wave schedule on the 128-step critical path, so it stays off
  constexpr bool SPLIT = false && (UB > 4) && (UB % 4 == 1) && (UB + 2 < NW);
  constexpr int HP = UB * 16, NP = UB * 64, KG = UB, OWN = (UB + NW - 1) / NW;
  const LstmProblem& P = a.p[blockIdx.z];
  const int H = a.H;
  const int s0 = blockIdx.x * 16;
  if (s0 >= P.n) return;
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, kk = lane >> 4;
  __shared__ __attribute__((aligned(16))) float hbuf[2][16 * HP];
  for (int i = tid; i < 2 * 16 * HP; i += NT) (&hbuf[0][0])[i] = 0.f;
  const int tmax = P.slen[s0];
  set_prio_by_length(tmax);
  const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
  if constexpr (SPLIT) {
    // ---- balanced schedule (UB = 4q + 1, e.g. H = 200 -> 13 blocks on 16 waves): waves 0..UB-2 own one unit block each
    // (4 gate chains = 4 * KG * 4 MFMAs); the LAST block's four gate chains go to waves UB-1..UB+2 (one chain each), so every
    // SIMD issues exactly (UB-1)/4 * 4 + 1 = UB chains per step instead of 16 vs 12.  The four chains meet through a 4 KB LDS
    // exchange; the split waves then update 4 rows each (one (row, unit) item per lane).
    __shared__ float zx[4 * 16 * 16];
    const bool full = w < UB - 1;
    const int ub = full ? w : UB - 1;
    const int pg = w - (UB - 1);
    const f32x4* wf = reinterpret_cast<const f32x4*>(P.wfrag) + ((long)(d * UB + ub) * 4 * KG) * 64 + lane;
    f32x4 bcur[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bcur[g] = wf[((full ? g : pg) * KG + 0) * 64];
    f32x4 x[4];
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = 0.f;
    // 32-bit element offsets off the scalar base pointers (the launcher checks rows * ld < 2^31): 64-bit per-lane
    // addresses would not fit the 128-VGPR budget of a 16-wave workgroup next to the MFMA operands
    auto load_x = [&](int step) __attribute__((always_inline)) {
      const int t = d ? (tmax - 1 - step) : step;
      const int nact = min(16, P.bs[t] - s0);
      const unsigned row0 = (unsigned)(P.off[t] + s0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = full ? kk * 4 + e : pg * 4 + kk;
        x[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((full || e == 0) && row < nact)
          x[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + ((row0 + row) * (unsigned)ldg + d * NP + ub * 64 + r * 4)));
      }
    };
    auto cell_update = [&](int row, int unit, unsigned row0, float zi, float zf, float zg, float zo, f32x4 xv, float cst, float* hn)
        __attribute__((always_inline)) -> float {
      const float gi = fast_sigmoid(zi + xv[0]);
      const float gf = fast_sigmoid(zf + xv[1]);
      const float gg = fast_tanh(zg + xv[2]);
      const float go = fast_sigmoid(zo + xv[3]);
      const float cn = gf * cst + gi * gg;
      const float hv = go * fast_tanh(cn);
      const unsigned rr = row0 + row;
      *reinterpret_cast<f32x4*>(P.gates + (rr * (unsigned)ldg + d * NP + (unit >> 4) * 64 + (unit & 15) * 4)) = f32x4{gi, gf, gg, go};
      P.cell[rr * (unsigned)ldc + d * HP + unit] = cn;
      if (unit < H) P.hout[rr * (unsigned)ldh + d * H + unit] = hv;
      hn[lds_off(row, unit, HP)] = hv;
      return cn;
    };
    __syncthreads();
    load_x(0);
    int cur = 0;
    for (int step = 0; step < tmax; ++step) {
      const int t = d ? (tmax - 1 - step) : step;
      const int nact = min(16, P.bs[t] - s0);
      const unsigned row0 = (unsigned)(P.off[t] + s0);
      const float* hc = hbuf[cur];
      float* hn = hbuf[cur ^ 1];
      if (full) {
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        // no explicit double buffer here: with 4 waves per SIMD a wave's fragment loads (L2 hits) hide under the other three
        // waves' 3 x 16 MFMAs, and the 16 VGPRs a second fragment set would cost push this path over the 128-register budget
        // (rolled loop on purpose: fully unrolled, hipcc hoists many fragment loads ahead and spills)
#pragma unroll 1
        for (int kg = 0; kg < KG; ++kg) {
          f32x4 b[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) b[g] = (kg == 0) ? bcur[g] : wf[(g * KG + kg) * 64];
          const f32x4 af = *reinterpret_cast<const f32x4*>(&hc[r * HP + kg * 16 + 4 * (kk ^ swz16(r))]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], b[g][i], acc[g], 0, 0, 0);
        }
        // the first fragments of the NEXT step load under this step's cell update and barriers
#pragma unroll
        for (int g = 0; g < 4; ++g) bcur[g] = wf[(g * KG + 0) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = kk * 4 + e;
          if (row < nact) c[e] = cell_update(row, ub * 16 + r, row0, acc[0][e], acc[1][e], acc[2][e], acc[3][e], x[e], c[e], hn);
        }
      } else {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
          const int kn = (kg + 1 < KG) ? kg + 1 : 0;
          const f32x4 bn = wf[(pg * KG + kn) * 64];
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 af = *reinterpret_cast<const f32x4*>(&hc[r * HP + kg * 16 + 4 * (kk ^ swz16(r))]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bcur[0][i], acc, 0, 0, 0);
          bcur[0] = bn;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) zx[(pg * 16 + kk * 4 + e) * 16 + r] = acc[e];
      }
      __syncthreads();
      if (!full) {
        const int row = pg * 4 + kk;
        if (row < nact)
          c[0] = cell_update(row, ub * 16 + r, row0, zx[(0 * 16 + row) * 16 + r], zx[(1 * 16 + row) * 16 + r], zx[(2 * 16 + row) * 16 + r],
                             zx[(3 * 16 + row) * 16 + r], x[0], c[0], hn);
      }
      if (step + 1 < tmax) load_x(step + 1);
      __syncthreads();
      cur ^= 1;
    }
    const int unit = ub * 16 + r;
    if (unit < H) {
      if (full) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int sq = s0 + kk * 4 + e;
          if (sq < P.n) P.cn[(long)sq * ldh + d * H + unit] = c[e];
        }
      } else {
        const int sq = s0 + pg * 4 + kk;
        if (sq < P.n) P.cn[(long)sq * ldh + d * H + unit] = c[0];
      }
    }
    return;
  }
  float c[OWN][4];
#pragma unroll
  for (int o = 0; o < OWN; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) c[o][e] = 0.f;
  __syncthreads();

  int cur = 0;
  // W_hh fragments are the same every step: the (kg+1) % KG prefetch wraps around, so the first fragments of the NEXT step
  // are already in flight while this step's cell update and barrier run.  (OWN == 1 for H = 200: one unit block per wave.)
  f32x4 bcur[OWN][4];
#pragma unroll
  for (int o = 0; o < OWN; ++o) {
    const int ub = w + NW * o;
    const f32x4* wf = reinterpret_cast<const f32x4*>(P.wfrag) + ((long)(d * UB + (ub < UB ? ub : 0)) * 4 * KG) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) bcur[o][g] = wf[(g * KG + 0) * 64];
  }
  auto load_x = [&](int step, f32x4 (&x)[OWN][4]) {
    const int t = d ? (tmax - 1 - step) : step;
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
      const int ub = w + NW * o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        x[o][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ub < UB && row < nact)
          x[o][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + (row0 + row) * ldg + d * NP + ub * 64 + r * 4));
      }
    }
  };
  f32x4 x[OWN][4];
  load_x(0, x);
  for (int step = 0; step < tmax; ++step) {
    const int t = d ? (tmax - 1 - step) : step;
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    const float* hc = hbuf[cur];
    float* hn = hbuf[cur ^ 1];
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
      const int ub = w + NW * o;
      if (ub < UB) {
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4* wf = reinterpret_cast<const f32x4*>(P.wfrag) + ((long)(d * UB + ub) * 4 * KG) * 64 + lane;
        f32x4 bnxt[4];
        if (!(a.dbg & 4))
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
          const int kn = (kg + 1 < KG) ? kg + 1 : 0;
#pragma unroll
          for (int g = 0; g < 4; ++g) bnxt[g] = wf[(g * KG + kn) * 64];
          // keep the prefetch where it is: hipcc otherwise sinks these loads next to their first use (one k-group
          // later), exposing an L2 round trip per fragment on the critical path of the recurrence
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 af = *reinterpret_cast<const f32x4*>(&hc[r * HP + kg * 16 + 4 * (kk ^ swz16(r))]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bcur[o][g][i], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) bcur[o][g] = bnxt[g];
        }
        // lane-local cell update: lane holds (row = kk*4+e, unit = ub*16 + r)
        const int unit = ub * 16 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = kk * 4 + e;
          if (row < nact) {
            const float gi = fast_sigmoid(acc[0][e] + x[o][e][0]);
            const float gf = fast_sigmoid(acc[1][e] + x[o][e][1]);
            const float gg = fast_tanh(acc[2][e] + x[o][e][2]);
            const float go = fast_sigmoid(acc[3][e] + x[o][e][3]);
            const float cn = gf * c[o][e] + gi * gg;
            const float hv = go * fast_tanh(cn);
            c[o][e] = cn;
            if (!(a.dbg & 1)) {
            // plain (write-back) stores: they retire from the in-order vmcnt queue at L2, so the next step's W_hh
            // fragment loads do not wait behind an HBM write (non-temporal stores cost +1.4 us per step here)
            *reinterpret_cast<f32x4*>(P.gates + (row0 + row) * ldg + d * NP + ub * 64 + r * 4) = f32x4{gi, gf, gg, go};
            P.cell[(row0 + row) * ldc + d * HP + unit] = cn;
            if (unit < H) P.hout[(row0 + row) * ldh + d * H + unit] = hv;
            }
            hn[lds_off(row, unit, HP)] = hv;
          }
        }
      }
    }
    if (step + 1 < tmax && !(a.dbg & 2)) load_x(step + 1, x);       // in flight across the barrier and the next step's MFMA loop
    if (!(a.dbg & 8)) __syncthreads();
    cur ^= 1;
  }
  // final cell state (forward: after t = len-1, reverse: after t = 0) -- rows keep c once they go inactive
#pragma unroll
  for (int o = 0; o < OWN; ++o) {
    const int ub = w + NW * o, unit = ub * 16 + r;
    if (ub < UB && unit < H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = s0 + kk * 4 + e;
        if (s < P.n) P.cn[(long)s * ldh + d * H + unit] = c[o][e];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Per step: (1) lane-local gate gradients from the saved activations -> dgates tile in LDS (+ global, in place over
// the saved gates);  (2) dh_{prev} = dgates[16, 4H] . W_hh on the matrix cores, W_hh streamed from L2 in fragment
// layout; dh_prev / dc_prev stay in registers in the same lane that needs them next step.
template <int UB>
__global__ __launch_bounds__((UB > 4 ? 16 : 4) * 64) void lstm_bwd_kernel(LstmArgs a) {
  constexpr int NW = UB > 4 ? 16 : 4;
  constexpr int HP = UB * 16, NP = UB * 64, KGB = NP / 16, OWN = (UB + NW - 1) / NW;
  const LstmProblem& P = a.p[blockIdx.z];
  const int H = a.H;
  const int s0 = blockIdx.x * 16;
  if (s0 >= P.n) return;
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, kk = lane >> 4;
  constexpr int DLD = NP + 16;             // row stride = 4 (mod 16) 16-byte chunks: the swizzle needs it (NP alone is 0 mod 16)
  __shared__ __attribute__((aligned(16))) float dg[16 * DLD];
  const int tmax = P.slen[s0];
  set_prio_by_length(tmax);
  int mylen[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int s = s0 + kk * 4 + e;
    mylen[e] = (s < P.n) ? P.slen[s] : 0;
  }
  float dhr[OWN][4], dcr[OWN][4];
#pragma unroll
  for (int o = 0; o < OWN; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dhr[o][e] = 0.f; dcr[o][e] = 0.f; }

  const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
  // software-pipelined inputs of the gate-gradient phase (loaded one step ahead, under the MFMA phase)
  f32x4 in_g[OWN][4];
  float in_ct[OWN][4], in_cp[OWN][4], in_dh[OWN][4];
  auto load_inputs = [&](int step) {
    const int t = d ? step : (tmax - 1 - step);
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    // previous step of the FORWARD recurrence: t-1 (forward dir) / t+1 (reverse dir, if inside the sequence)
    const int tp = d ? t + 1 : t - 1;
    const long prow0 = (tp >= 0 && tp < P.L) ? (long)P.off[tp] + s0 : 0;
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
      const int ub = w + NW * o;
      if (ub < UB) {
        const int unit = ub * 16 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = kk * 4 + e;
          in_g[o][e] = f32x4{0.f, 0.f, 0.f, 0.f};
          in_ct[o][e] = 0.f; in_cp[o][e] = 0.f; in_dh[o][e] = 0.f;
          if (row < nact) {
            const long grow = row0 + row;
            in_g[o][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + grow * ldg + d * NP + ub * 64 + r * 4));
            in_ct[o][e] = __builtin_nontemporal_load(P.cell + grow * ldc + d * HP + unit);
            const bool has_prev = d ? (t + 1 < mylen[e]) : (t > 0);
            if (has_prev) in_cp[o][e] = __builtin_nontemporal_load(P.cell + (prow0 + row) * ldc + d * HP + unit);
            if (unit < H) in_dh[o][e] = __builtin_nontemporal_load(P.dh + grow * ldh + d * H + unit);
          }
        }
      }
    }
  };
  load_inputs(0);
  constexpr int PF = 4;
  static_assert(KGB % PF == 0, "prefetch ring must divide the fragment count");
  f32x4 ring[OWN][PF];
#pragma unroll
  for (int o = 0; o < OWN; ++o) {
    const int ub = w + NW * o;
#pragma unroll
    for (int j = 0; j < PF; ++j)
      ring[o][j] = (ub < UB) ? (reinterpret_cast<const f32x4*>(P.wfrag) + (long)d * UB * KGB * 64 + lane)[((long)ub * KGB + j) * 64]
                             : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int step = 0; step < tmax; ++step) {
    const int t = d ? step : (tmax - 1 - step);          // reverse of the forward pass's order
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    // ---- (1) gate gradients
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
      const int ub = w + NW * o;
      if (ub < UB) {
        const int unit = ub * 16 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = kk * 4 + e;
          f32x4 dgv = {0.f, 0.f, 0.f, 0.f};
          if (row < nact) {
            const long grow = row0 + row;
            const float gi = in_g[o][e][0], gf = in_g[o][e][1], gg = in_g[o][e][2], go = in_g[o][e][3];
            const float ct = in_ct[o][e], cp = in_cp[o][e];
            const float dh = dhr[o][e] + in_dh[o][e];
            float dc = dcr[o][e];
            // first step of this row's backward = last step of its forward: add dL/dc_n
            const bool last_fwd_step = d ? (t == 0) : (t == mylen[e] - 1);
            if (last_fwd_step && P.dcn && unit < H) dc += P.dcn[(long)(s0 + row) * ldh + d * H + unit];
            const float tc = fast_tanh(ct);
            dgv[3] = dh * tc * go * (1.f - go);
            dc += dh * go * (1.f - tc * tc);
            dgv[0] = dc * gg * gi * (1.f - gi);
            dgv[1] = dc * cp * gf * (1.f - gf);
            dgv[2] = dc * gi * (1.f - gg * gg);
            dcr[o][e] = dc * gf;
            *reinterpret_cast<f32x4*>(P.gates + grow * ldg + d * NP + ub * 64 + r * 4) = dgv;
          }
          *reinterpret_cast<f32x4*>(&dg[lds_off(row, ub * 64 + r * 4, DLD)]) = dgv;
        }
      }
    }
    __syncthreads();
    if (step + 1 < tmax) load_inputs(step + 1);
    // ---- (2) dh_prev[16, HP] = dgates[16, NP] . W_hh (p-order rows)
    {
      f32x4 acc[OWN];
#pragma unroll
      for (int o = 0; o < OWN; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4* wb = reinterpret_cast<const f32x4*>(P.wfrag) + (long)d * UB * KGB * 64 + lane;
      // ring of PF fragments in flight: one fragment feeds only 4 MFMAs (128 cycles) here, far less than an L2 round trip
#pragma unroll 4
      for (int kg = 0; kg < KGB; ++kg) {
        const int kn = (kg + PF < KGB) ? kg + PF : kg + PF - KGB;     // wraps into the next step
        const f32x4 af = *reinterpret_cast<const f32x4*>(&dg[r * DLD + kg * 16 + 4 * (kk ^ swz16(r))]);
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
          if (w + NW * o < UB) {
            const f32x4 b = ring[o][kg % PF];
            ring[o][kg % PF] = wb[((long)(w + NW * o) * KGB + kn) * 64];
            __builtin_amdgcn_sched_barrier(0);      // pin the refill ahead of the MFMAs (see the forward kernel)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], b[i], acc[o], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < OWN; ++o)
#pragma unroll
        for (int e = 0; e < 4; ++e) dhr[o][e] = acc[o][e];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ weight (un)packing
__global__ void lstm_pack_kernel(const float* __restrict__ w_ih_f, const float* __restrict__ w_hh_f,
                                 const float* __restrict__ b_ih_f, const float* __restrict__ b_hh_f,
                                 const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                 const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r, int H, int E, int UB,
                                 float* __restrict__ w_ihp, float* __restrict__ b_p, float* __restrict__ wf,
                                 float* __restrict__ wb) {
  const int NP = UB * 64, KG = UB, KGB = NP / 16;
  const long n_ihp = (long)2 * NP * E, n_b = 2 * NP, n_wf = (long)2 * UB * 4 * KG * 256, n_wb = (long)2 * UB * KGB * 256;
  const long total = n_ihp + n_b + n_wf + n_wb;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long i = idx;
    if (i < n_ihp) {                       // w_ihp[d*NP + p][e]
      const int e = i % E; const int dp = i / E; const int d = dp / NP, p = dp % NP;
      const int ub = p / 64, u = (p % 64) / 4, g = p % 4, unit = ub * 16 + u;
      const float* src = d ? w_ih_r : w_ih_f;
      w_ihp[i] = (unit < H) ? src[(long)(g * H + unit) * E + e] : 0.f;
      continue;
    }
    i -= n_ihp;
    if (i < n_b) {
      const int d = i / NP, p = i % NP;
      const int ub = p / 64, u = (p % 64) / 4, g = p % 4, unit = ub * 16 + u;
      const float* bi = d ? b_ih_r : b_ih_f; const float* bh = d ? b_hh_r : b_hh_f;
      b_p[i] = (unit < H) ? bi[g * H + unit] + bh[g * H + unit] : 0.f;
      continue;
    }
    i -= n_b;
    if (i < n_wf) {                        // wf[d][ub][g][kg][lane][ii] = w_hh[g*H + ub*16 + (lane&15)][16kg + 4(lane>>4) + ii]
      const int ii = i & 3, lane = (i >> 2) & 63; long q = i >> 8;
      const int kg = q % KG; q /= KG; const int g = q % 4; q /= 4; const int ub = q % UB; const int d = q / UB;
      const int unit = ub * 16 + (lane & 15), k = 16 * kg + 4 * (lane >> 4) + ii;
      const float* src = d ? w_hh_r : w_hh_f;
      wf[i] = (unit < H && k < H) ? src[(long)(g * H + unit) * H + k] : 0.f;
      continue;
    }
    i -= n_wf;
    {                                      // wb[d][ubn][kg][lane][ii] = w_hh[row(p)][ubn*16 + (lane&15)],  p = 16kg + 4(lane>>4) + ii
      const int ii = i & 3, lane = (i >> 2) & 63; long q = i >> 8;
      const int kg = q % KGB; q /= KGB; const int ubn = q % UB; const int d = q / UB;
      const int p = 16 * kg + 4 * (lane >> 4) + ii;
      const int ub = p / 64, u = (p % 64) / 4, g = p % 4, unit = ub * 16 + u, col = ubn * 16 + (lane & 15);
      const float* src = d ? w_hh_r : w_hh_f;
      wb[i] = (unit < H && col < H) ? src[(long)(g * H + unit) * H + col] : 0.f;
    }
  }
}

// dW_ihp [2*NP, E], db_p [2*NP], dW_hhp [2][NP][H]  ->  reference-layout gradients (overwrite)
__global__ void lstm_unpack_kernel(const float* __restrict__ dw_ihp, const float* __restrict__ db_p,
                                   const float* __restrict__ dw_hhp, int H, int E, int UB, float* __restrict__ dw_ih_f,
                                   float* __restrict__ dw_hh_f, float* __restrict__ db_ih_f, float* __restrict__ db_hh_f,
                                   float* __restrict__ dw_ih_r, float* __restrict__ dw_hh_r, float* __restrict__ db_ih_r,
                                   float* __restrict__ db_hh_r) {
  const int NP = UB * 64;
  const long n_ih = (long)2 * 4 * H * E, n_hh = (long)2 * 4 * H * H, n_b = 2 * 4 * H;
  const long total = n_ih + n_hh + n_b;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long i = idx;
    if (i < n_ih) {
      const int e = i % E; long q = i / E; const int row = q % (4 * H); const int d = q / (4 * H);
      const int g = row / H, unit = row % H, p = (unit / 16) * 64 + (unit % 16) * 4 + g;
      (d ? dw_ih_r : dw_ih_f)[(long)row * E + e] = dw_ihp[(long)(d * NP + p) * E + e];
      continue;
    }
    i -= n_ih;
    if (i < n_hh) {
      const int k = i % H; long q = i / H; const int row = q % (4 * H); const int d = q / (4 * H);
      const int g = row / H, unit = row % H, p = (unit / 16) * 64 + (unit % 16) * 4 + g;
      (d ? dw_hh_r : dw_hh_f)[(long)row * H + k] = dw_hhp[((long)d * NP + p) * H + k];
      continue;
    }
    i -= n_hh;
    {
      const int row = i % (4 * H), d = i / (4 * H);
      const int g = row / H, unit = row % H, p = (unit / 16) * 64 + (unit % 16) * 4 + g;
      const float v = db_p[d * NP + p];
      (d ? db_ih_r : db_ih_f)[row] = v;
      (d ? db_hh_r : db_hh_f)[row] = v;
    }
  }
}

template <int UB>
int launch_rec(const LstmArgs& a, bool backward, int max_tiles, hipStream_t s) {
  dim3 grid(max_tiles, 2, a.nprob), block((UB > 4 ? 16 : 4) * 64);
  if (backward) hipLaunchKernelGGL((lstm_bwd_kernel<UB>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((lstm_fwd_kernel<UB>), grid, block, 0, s, a);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

}  // namespace

extern "C" int nnr_lstm_dims(int H, int* UB, int* HP, int* NP) {
  const int ub = (H + 15) / 16;
  if (UB) *UB = ub;
  if (HP) *HP = ub * 16;
  if (NP) *NP = ub * 64;
  return (ub == 1 || ub == 2 || ub == 13) ? NNR_OK : NNR_ERR_UNSUPPORTED;
}

extern "C" int nnr_lstm_pack_weights(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                                     const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                     int H, int E, float* w_ihp, float* b_p, float* wf, float* wb, hipStream_t stream) {
  int UB;
  if (nnr_lstm_dims(H, &UB, nullptr, nullptr) != NNR_OK) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(1024), dim3(256), 0, stream, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r,
                     b_ih_r, b_hh_r, H, E, UB, w_ihp, b_p, wf, wb);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_lstm_unpack_grads(const float* dw_ihp, const float* db_p, const float* dw_hhp, int H, int E,
                                     float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f, float* dw_ih_r,
                                     float* dw_hh_r, float* db_ih_r, float* db_hh_r, hipStream_t stream) {
  int UB;
  if (nnr_lstm_dims(H, &UB, nullptr, nullptr) != NNR_OK) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lstm_unpack_kernel, dim3(1024), dim3(256), 0, stream, dw_ihp, db_p, dw_hhp, H, E, UB, dw_ih_f,
                     dw_hh_f, db_ih_f, db_hh_f, dw_ih_r, dw_hh_r, db_ih_r, db_hh_r);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

static int lstm_run(const nnr_lstm_problem* probs, int nprob, int H, bool backward, hipStream_t stream) {
  if (!probs || nprob < 1 || nprob > 4) return NNR_ERR_ARG;
  int UB;
  if (nnr_lstm_dims(H, &UB, nullptr, nullptr) != NNR_OK) return NNR_ERR_UNSUPPORTED;
  LstmArgs a;
  a.nprob = nprob;
  a.H = H;
  { const char* e = getenv("NNR_LSTM_DBG"); a.dbg = e ? atoi(e) : 0; }
  int max_tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const nnr_lstm_problem& q = probs[i];
    LstmProblem& p = a.p[i];
    p.bs = q.bs; p.off = q.off; p.slen = q.slen; p.prev_f = q.prev_f; p.prev_r = q.prev_r;
    p.n = q.n; p.L = q.L;
    p.gates = q.gates; p.cell = q.cell; p.hout = q.hout; p.cn = q.cn;
    p.wfrag = backward ? q.wb : q.wf;
    p.dh = q.dh; p.dcn = q.dcn;
    if (!p.bs || !p.off || !p.slen || !p.gates || !p.cell || !p.wfrag || p.n <= 0) return NNR_ERR_ARG;
    if (backward ? (!p.dh || !p.prev_f || !p.prev_r) : (!p.hout || !p.cn)) return NNR_ERR_ARG;
    max_tiles = max(max_tiles, (p.n + 15) / 16);
  }
  for (int i = nprob; i < 4; ++i) a.p[i] = a.p[0];
  switch (UB) {
    case 1: return launch_rec<1>(a, backward, max_tiles, stream);
    case 2: return launch_rec<2>(a, backward, max_tiles, stream);
    case 13: return launch_rec<13>(a, backward, max_tiles, stream);
  }
  return NNR_ERR_UNSUPPORTED;
}

extern "C" int nnr_lstm_fwd(const nnr_lstm_problem* probs, int nprob, int H, hipStream_t stream) {
  return lstm_run(probs, nprob, H, false, stream);
}
extern "C" int nnr_lstm_bwd(const nnr_lstm_problem* probs, int nprob, int H, hipStream_t stream) {
  return lstm_run(probs, nprob, H, true, stream);
}
