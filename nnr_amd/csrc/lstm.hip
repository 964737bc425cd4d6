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
  unsigned* sync;      // pair kernels: [2][ntiles][2][SYNC_PAD] step counters + [SYNC_PAD] diagnostics
};
constexpr int SYNC_PAD = 16;                 // diagnostics block (64 bytes) at the end of the workspace
constexpr int SPIN_LIMIT = 1 << 17;          // ~0.2 s: a partner workgroup may still be waiting for a free CU (that wait is bounded by
                                             // the kernels running next to this one: milliseconds)

// Exchange between two RUNNING workgroups.  Every value travels as one 8-byte (value, step tag) word, so the payload
// carries its own readiness: no flag, no fence, no store-acknowledge wait -- the reader polls the words it needs.
// Two flavours, chosen per pair at kernel start from the XCC_ID hardware register of both workgroups:
//  * same XCD (the normal case: workgroups i and i + 8 of a dispatch land on the same XCD, 128/128 pairs in
//    tools/micro/xcdpp.hip): plain stores (the per-CU cache is write-through) + NON-TEMPORAL loads, which never retain a
//    line in the per-CU cache and therefore always read the XCD's L2 -- 1.2 us per full exchange round, 256 workgroups
//    at once, 0 stale words in 300 x 128 x 1792 checked;
//  * different XCDs (their L2s are not coherent with each other): relaxed agent-scope atomics = write-through stores /
//    cache-bypassing loads (sc1) through the fabric -- 2.1-3.3 us per round.  (Agent-scope release/acquire FENCES cost
//    14-19 us per round at that occupancy, tools/micro/pingpong.hip; group-scope sc0 loads hit stale per-CU lines.)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ void st_tag(unsigned long long* p, float v, unsigned tag, bool same_xcd) {
  const unsigned long long word = ((unsigned long long)tag << 32) | __float_as_uint(v);
  if (same_xcd) __hip_atomic_store(p, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // a plain store: no wait, no cache-policy bits
  else __hip_atomic_store(p, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_tag(const unsigned long long* p, bool same_xcd) {
  if (same_xcd) {
    unsigned long long v;      // asm: a polling loop must re-issue the load every time
    asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// first (batched) read of a step: compiler-tracked, so several can be in flight
__device__ __forceinline__ unsigned long long ld_tag_first(const unsigned long long* p, bool same_xcd) {
  if (same_xcd) return __builtin_nontemporal_load(p);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// re-poll one word until its tag matches (bounded: never hang the GPU -- on timeout count it and return NaN)
__device__ __forceinline__ float wait_tag(const unsigned long long* p, unsigned long long v, unsigned tag, bool same_xcd, unsigned* diag,
                                          unsigned* total) {
  int spins = 0;
  while ((unsigned)(v >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    v = ld_tag(p, same_xcd);
    // the partner never delivered (or another wait of this launch already gave up: do not stack timeouts): fail LOUDLY --
    // count it, and poison the value so the step's loss becomes NaN instead of silently training on stale data
    if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(diag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicAdd(diag, 1u);
      if (total) atomicAdd(total, 1u);      // persistent across launches (nnr_lstm_set_timeout_counter): a time-out is a DATA-POISONING
      v = 0x7fc00000ull;                    // event (NaN), not a retry -- the optimizer skips a step whose gradient norm is not finite
      break;
    }
  }
#ifdef NNR_LSTM_COUNT_REPOLLS
  if (spins) { atomicAdd(diag + 1, 1u); atomicAdd(diag + 2, (unsigned)spins); }
#endif
  return __uint_as_float((unsigned)v);
}

struct LstmArgs { LstmProblem p[4]; int nprob; int H; int dbg; unsigned* tmo_total; int quad_T; unsigned epoch; };   // epoch: launch counter << 10, the high bits of every exchange tag   // dbg: timing-attribution mask (NNR_LSTM_DBG), 0 in production

// ------------------------------------------------------------------------------------------------ forward
template <int UB>
__global__ __launch_bounds__((UB > 4 ? 16 : 4) * 64) void lstm_fwd_kernel(LstmArgs a) {
  constexpr int NW = UB > 4 ? 16 : 4, NT = NW * 64;      // one 16-unit block per wave when the hidden size is large
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


// ------------------------------------------------------------------------------------------------ forward, 2 CUs per tile
// Weights-stationary variant: a tile's gate columns are split over a PAIR of workgroups (pair_id / pair_half) on two CUs
// of one XCD.  Half 0 owns unit blocks [0, UB0), half 1 owns [UB0, UB).  Each of a workgroup's compute waves keeps the W_hh
// fragments of its unit block RESIDENT for the whole sequence -- 3 gates in registers (3*KG f32x4 = 156 VGPRs at H = 200),
// the 4th in LDS -- so nothing is streamed from L2 inside the time loop (the one-CU kernel re-reads 640 KB per step), and
// the per-step MFMA chain per SIMD is halved.  The price is one exchange per step: each half needs the other half's units
// of h_{t-1}.  h_t is written to `hout` anyway; it is written with write-through stores, a per-workgroup step counter is
// published, and a helper wave of the partner copies the rows into its LDS tile with cache-bypassing loads while the
// compute waves are busy with the K range they own (phase A); the partner's K range follows after a barrier (phase B).
// Dispatch is in blockIdx order, so a waiting workgroup's partner is always the next one to get a CU: no deadlock; the
// spin loops are bounded anyway.
// Pairing: workgroups x and x + 8 of a dispatch share an XCD (round-robin placement), so within every group of 16
// consecutive workgroups the first 8 are halves 0 and the next 8 the matching halves 1 of 8 tiles.
// Dispatch order = longest first over ALL token streams and directions (LPT): the 1-D grid walks groups of 16 workgroups
// (8 tiles x 2 halves); group G holds tiles [8g, 8g + 8) of stream (problem, direction) = G % (2 nprob), g = G / (2 nprob).  (A
// (tiles, direction, problem) grid dispatched every tile of direction 0 before the 128-step tiles of direction 1 got a CU:
// 1.19 ms for a 1.0 ms chain.)
__device__ __forceinline__ int pair_half(int bx) { return (bx >> 3) & 1; }
struct PairId { int prob, d, tile; };
__device__ __forceinline__ PairId pair_id(int nprob) {
  const int bx = blockIdx.x, G = bx >> 4, per = 2 * nprob;
  const int g = G / per, rem = G - g * per;
  return PairId{rem >> 1, rem & 1, g * 8 + (bx & 7)};
}

// QUAD tiles.  A tile's chain of dependent steps is bound by the MFMA work of a step (5.6 of 8.6 us with 16 rows), and the launch by
// its longest chain (128 steps), not by throughput.  Sequences longer than quad_T steps are therefore run in tiles of FOUR rows on
// v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 rows x 4 columns x 1 k per instruction, same FLOP rate, a quarter of the work per
// step: tools/micro/mfma4.hip checks the operand layout and the 8.6-clock issue rate): ~3 us per step instead of 8.6 / 9.8.
// Sorted positions [0, 4 nq) are quad tiles (nq from the plan's batch sizes, on the device: bs[T] sequences are longer than T),
// positions from 4 nq on are 16-row tiles.  Tile ids: quads 0 .. nq-1 (dispatched first: longest), 16-row tiles nq + j.
constexpr int MAXQ = 64;
__device__ __forceinline__ int quad_count(const LstmArgs& a, const LstmProblem& P) {
  const int T = a.quad_T;
  if (T <= 0 || T >= P.L) return 0;
  return min((P.bs[T] + 3) >> 2, MAXQ);
}
__host__ __device__ inline int sync_tiles(int n) { return (n + 15) / 16 + MAXQ; }     // exchange slots per (stream, direction)
#define MFMA4_BCAST(a, b, c, abid) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (abid), 0)   /* A of block `abid` for all 16 blocks */
#define MFMA4_OWN(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)                /* every block its own A */
__device__ __forceinline__ float quad_bcast(float v, int g) {     // value of lane (quad, g) in all four lanes of the quad
  const int x = __float_as_int(v);
  int r;
  switch (g) {
    case 0: r = __builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true); break;
    case 1: r = __builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true); break;
    case 2: r = __builtin_amdgcn_mov_dpp(x, 0xaa, 0xf, 0xf, true); break;
    default: r = __builtin_amdgcn_mov_dpp(x, 0xff, 0xf, 0xf, true); break;
  }
  return __int_as_float(r);
}

template <int UB, int hv>
__device__ __forceinline__ void lstm_fwd_pair_body(const LstmArgs& a, const LstmProblem& P, const int d, const int tile, const int s0,
                                                   float (*hbuf)[16 * UB * 16], f32x4 (*wl)[UB][64]) {
  constexpr int HP = UB * 16, NP = UB * 64, KG = UB, UB0 = (UB + 1) / 2, NW = 8, NT = NW * 64;
  const int H = a.H;
  if (s0 >= P.n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, kk = lane >> 4;
  constexpr int ub_lo = hv ? UB0 : 0, nb = hv ? UB - UB0 : UB0;
  const int tmax = P.slen[s0];
  if (tmax == 1) {
    // Every sequence of the tile is ONE token long (padded history slots: all-zero text, mask[:,0] = 1 -- half of the 3 520
    // sequences of a MIND-shaped batch): h_0 = c_0 = 0, so z = xw and the step is element-wise.  No weights, no partner, no
    // exchange: such a tile cost ~24 us of start-up (weight fragments, placement handshake) for one 9 us step.
    if (w < nb) {
      const int ub = ub_lo + w, unit = ub * 16 + r;
      const int nact = min(16, P.bs[0] - s0);
      const long row0 = (long)P.off[0] + s0;
      const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        if (row < nact) {
          float* gp = P.gates + (row0 + row) * ldg + d * NP + ub * 64 + r * 4;
          const f32x4 x = *reinterpret_cast<const f32x4*>(gp);
          const float gi = fast_sigmoid(x[0]), gf = fast_sigmoid(x[1]), gg = fast_tanh(x[2]), go = fast_sigmoid(x[3]);
          const float cn = gi * gg, hval = go * fast_tanh(cn);
          *reinterpret_cast<f32x4*>(gp) = f32x4{gi, gf, gg, go};
          P.cell[(row0 + row) * ldc + d * HP + unit] = cn;
          if (unit < H) {
            P.hout[(row0 + row) * ldh + d * H + unit] = hval;
            P.cn[(long)(s0 + row) * ldh + d * H + unit] = cn;
          }
        }
      }
    }
    return;
  }
  const int ntiles = sync_tiles(P.n);
  // exchange slots: [d][tile][half][step parity][16 rows][XW] tagged words, written by `half`, read by its partner
  constexpr int XW = UB0 * 16, XT = 16 * XW, XS = 2 * XT + 8;        // + one line for the placement handshake
  unsigned long long* xmine = reinterpret_cast<unsigned long long*>(P.sync) + ((long)(d * ntiles + tile) * 2 + hv) * XS;
  const unsigned long long* xtheirs = reinterpret_cast<const unsigned long long*>(P.sync) + ((long)(d * ntiles + tile) * 2 + (hv ^ 1)) * XS;
  unsigned* diag = P.sync + (long)2 * ntiles * 2 * XS * 2;
  // placement handshake through the fabric path (valid wherever the partner runs): do both halves sit on one XCD?
  if (tid == 0) {
    const unsigned mine = xcc_id();
    st_tag(xmine + 2 * XT, __uint_as_float(mine), a.epoch + 0x3ffu, false);
    const float theirs = wait_tag(xtheirs + 2 * XT, ld_tag(xtheirs + 2 * XT, false), a.epoch + 0x3ffu, false, diag, a.tmo_total);
    hbuf[0][0] = (__float_as_uint(theirs) == mine && !(a.dbg & 64)) ? 1.f : 0.f;
  }
  __syncthreads();
  const bool same_xcd = __builtin_amdgcn_readfirstlane(hbuf[0][0] != 0.f);
  __syncthreads();
  for (int i = tid; i < 2 * 16 * HP; i += NT) (&hbuf[0][0])[i] = 0.f;
  unsigned long long* tbuf = reinterpret_cast<unsigned long long*>(diag + SYNC_PAD);
  const bool stamp = (a.dbg & 32) && blockIdx.x == 0 && tid == 0;
#define STAMP(slot) do { if (stamp && step < 128) tbuf[step * 16 + (slot)] = wall_clock64(); } while (0)

  // the partner's REAL units (its padded ones stay 0 in LDS)
  const int pu0 = hv ? 0 : UB0 * 16, pw = (hv ? UB0 * 16 : H) - pu0;
  const bool compute = w < nb;
  const int ub = ub_lo + (compute ? w : 0);
  const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
  // resident weights: gates i, f, g in registers, gate o in LDS
  f32x4 wr[3][KG];
  {
    const f32x4* wf = reinterpret_cast<const f32x4*>(P.wfrag) + ((long)(d * UB + ub) * 4 * KG) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) wr[g][kg] = wf[(g * KG + kg) * 64];
    if (compute) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) wl[w][kg][lane] = wf[(3 * KG + kg) * 64];
    }
  }
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int step = 0; step < tmax; ++step) {
    const int t = d ? (tmax - 1 - step) : step;
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    float* hc = hbuf[step & 1];
    float* hn = hbuf[(step & 1) ^ 1];
    f32x4 acc[4];
    STAMP(4);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // every wave fetches its share of the partner's half of h_{t-1}: on either path a wave sustains only a few of these
    // requests (one wave fetching the whole tile takes 5.5 us through L2 and ~10 us through the fabric, eight waves sharing
    // it ~1 us; tools/micro/pingpong.hip).  The loads are
    // issued part-way through phase A -- late enough that the partner's stores (sent ~1 us before this step began, visible
    // ~2 us after being sent) have landed, early enough that the round trip hides under the remaining MFMAs.
    constexpr int MAXW = (XT + NT - 1) / NT;
    constexpr int KSPLIT = 3;                               // own k-groups done before the fetch is issued
    unsigned long long v[MAXW];
    const int tp = d ? t + 1 : t - 1;
    const int nprev = step > 0 ? min(16, P.bs[tp] - s0) : 0;
    const unsigned long long* src = xtheirs + ((step + 1) & 1) * XT;
    auto phase_a = [&](int k_from, int k_to) __attribute__((always_inline)) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const int own = kg - (hv ? UB0 : 0);                // index inside the own K range
        if ((kg < UB0) == (hv == 0) && own >= k_from && own < k_to) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(&hc[r * HP + kg * 16 + 4 * (kk ^ swz16(r))]);
          const f32x4 b3 = wl[w][kg][lane];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wr[g][kg][i], acc[g], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], b3[i], acc[3], 0, 0, 0);
          }
        }
      }
    };
    if (compute && !(a.dbg & 16)) phase_a(0, KSPLIT);
    __builtin_amdgcn_sched_barrier(0);
    if (!(a.dbg & 1)) {
#pragma unroll
      for (int j = 0; j < MAXW; ++j) {
        const int i = tid + NT * j, row = i / XW, cc = i - row * XW;
        v[j] = 0;
        if (row < nprev && cc < pw) v[j] = ld_tag_first(src + i, same_xcd);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (compute && !(a.dbg & 16)) phase_a(KSPLIT, KG);
    STAMP(5);
    if (!(a.dbg & 1)) {
      const unsigned tag = a.epoch + (unsigned)step;
#pragma unroll
      for (int j = 0; j < MAXW; ++j) {
        const int i = tid + NT * j, row = i / XW, cc = i - row * XW;
        if (row < nprev && cc < pw) hc[lds_off(row, pu0 + cc, HP)] = wait_tag(src + i, v[j], tag, same_xcd, diag, a.tmo_total);
      }
    }
    STAMP(2);
    __syncthreads();                                     // the partner's half of h_{t-1} is in LDS
    STAMP(6);
    if (compute) {
      f32x4 x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        x[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < nact) x[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + (row0 + row) * ldg + d * NP + ub * 64 + r * 4));
      }
      if (!(a.dbg & 8)) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        if ((kg < UB0) != (hv == 0)) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(&hc[r * HP + kg * 16 + 4 * (kk ^ swz16(r))]);
          const f32x4 b3 = wl[w][kg][lane];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wr[g][kg][i], acc[g], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], b3[i], acc[3], 0, 0, 0);
          }
        }
      }
      }
      if (stamp && step < 128) tbuf[step * 16 + 7] = wall_clock64() + (unsigned long long)(acc[0][0] == 123.456f);
      const int unit = ub * 16 + r;
      float hv_[4], cn_[4];
      f32x4 gt[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gi = fast_sigmoid(acc[0][e] + x[e][0]);
        const float gf = fast_sigmoid(acc[1][e] + x[e][1]);
        const float gg = fast_tanh(acc[2][e] + x[e][2]);
        const float go = fast_sigmoid(acc[3][e] + x[e][3]);
        cn_[e] = gf * c[e] + gi * gg;
        hv_[e] = go * fast_tanh(cn_[e]);
        gt[e] = f32x4{gi, gf, gg, go};
      }
      if (stamp && step < 128) tbuf[step * 16 + 9] = wall_clock64() + (unsigned long long)(hv_[0] == 123.456f);
      // the partner waits for these: send them first
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        if (row < nact && unit < H && !(a.dbg & 2)) st_tag(xmine + (step & 1) * XT + row * XW + (unit - ub_lo * 16), hv_[e], a.epoch + (unsigned)(step + 1), same_xcd);
      }
      STAMP(10);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        if (row < nact) {
          c[e] = cn_[e];
          hn[lds_off(row, unit, HP)] = hv_[e];
          if (a.dbg & 4) continue;
          if (unit < H) P.hout[(row0 + row) * ldh + d * H + unit] = hv_[e];
          *reinterpret_cast<f32x4*>(P.gates + (row0 + row) * ldg + d * NP + ub * 64 + r * 4) = gt[e];
          P.cell[(row0 + row) * ldc + d * HP + unit] = cn_[e];
        }
      }
    }
    STAMP(8);
    __syncthreads();
  }
#undef STAMP
  if (compute) {
    const int unit = ub * 16 + r;
    if (unit < H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = s0 + kk * 4 + e;
        if (s < P.n) P.cn[(long)s * ldh + d * H + unit] = c[e];
      }
    }
  }
}

// ---- forward, quad tile (4 sequences) on a CU pair.  Same split of the unit blocks, same exchange slots (rows 0..3 of a
// 16-row slot) as the 16-row body.  A compute wave owns the 64 gate columns of its unit block, one column per lane
// (lane = 4 * unit + gate): per k one v_mfma_f32_4x4x1 with h_{t-1}[4 rows][k] broadcast from one block (ABID) -- a single
// ds_read_b32 of the k-major h tile (hq[k][row]) feeds 16 instructions.  W_hh: 156 of the 208 k in registers, 52 in LDS
// (the budget of the 16-row body).  After the MFMAs a lane holds ONE gate of four rows; a 4 x 4 transpose inside each quad of
// lanes (DPP) gives lane (unit, row) the four gates of its (row, unit): the cell update stays lane-local, one cell per lane.
template <int UB, int hv>
__device__ __forceinline__ void lstm_fwd_quad_body(const LstmArgs& a, const LstmProblem& P, const int d, const int q, float* hq, float* wlq) {
  constexpr int HP = UB * 16, NP = UB * 64, KG = UB, UB0 = (UB + 1) / 2, NT = 512;
  constexpr int RK = 12 * KG, QL = HP - RK;              // k < RK: W_hh column in registers; the rest in LDS
  static_assert(QL * 64 * 4 * UB0 <= (int)sizeof(f32x4) * UB0 * UB * 64, "quad weights must fit the 16-row body's LDS array");
  const int H = a.H;
  const int s0 = q * 4;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int ub_lo = hv ? UB0 : 0, nb = hv ? UB - UB0 : UB0;
  const int tmax = P.slen[s0];
  __builtin_amdgcn_s_setprio(3);
  const int ntl = sync_tiles(P.n);
  constexpr int XW = UB0 * 16, XT = 16 * XW, XS = 2 * XT + 8;
  unsigned long long* xmine = reinterpret_cast<unsigned long long*>(P.sync) + ((long)(d * ntl + q) * 2 + hv) * XS;
  const unsigned long long* xtheirs = reinterpret_cast<const unsigned long long*>(P.sync) + ((long)(d * ntl + q) * 2 + (hv ^ 1)) * XS;
  unsigned* diag = P.sync + (long)2 * ntl * 2 * XS * 2;
  if (tid == 0) {
    const unsigned mine = xcc_id();
    st_tag(xmine + 2 * XT, __uint_as_float(mine), a.epoch + 0x3ffu, false);
    const float theirs = wait_tag(xtheirs + 2 * XT, ld_tag(xtheirs + 2 * XT, false), a.epoch + 0x3ffu, false, diag, a.tmo_total);
    hq[0] = (__float_as_uint(theirs) == mine && !(a.dbg & 64)) ? 1.f : 0.f;
  }
  __syncthreads();
  const bool same_xcd = __builtin_amdgcn_readfirstlane(hq[0] != 0.f);
  __syncthreads();
  for (int i = tid; i < 2 * HP * 4; i += NT) hq[i] = 0.f;
  const int pu0 = hv ? 0 : UB0 * 16, pw = (hv ? UB0 * 16 : H) - pu0;     // the partner's real units
  const bool compute = w < nb;
  const int ub = ub_lo + (compute ? w : 0);
  const int u = lane >> 2, j = lane & 3;                  // MFMA: column (unit u, gate j); cell update: (row j, unit u)
  const int unit = ub * 16 + u;
  const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
  float wr[RK];
  {
    // wf[d][ub][gate][kg][lane' = unit + 16 * ((k >> 2) & 3)][k & 3] (lstm_pack_kernel's fragment layout of the 16-row kernels)
    const float* wf = P.wfrag + (((long)(d * UB + ub) * 4 + j) * KG) * 256 + u * 4;
#pragma unroll
    for (int k = 0; k < RK; ++k) wr[k] = wf[(k >> 4) * 256 + ((k >> 2) & 3) * 64 + (k & 3)];
    if (compute) {
#pragma unroll 4
      for (int k = RK; k < HP; ++k) wlq[((long)w * QL + (k - RK)) * 64 + lane] = wf[(k >> 4) * 256 + ((k >> 2) & 3) * 64 + (k & 3)];
    }
  }
  float c = 0.f;
  __syncthreads();

  for (int step = 0; step < tmax; ++step) {
    const int t = d ? (tmax - 1 - step) : step;
    const int nact = min(4, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    const float* hc = hq + (step & 1) * (HP * 4);
    float* hcw = hq + (step & 1) * (HP * 4);
    float* hn = hq + ((step & 1) ^ 1) * (HP * 4);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tp = d ? t + 1 : t - 1;
    const int nprev = step > 0 ? min(4, P.bs[tp] - s0) : 0;
    const unsigned long long* src = xtheirs + ((step + 1) & 1) * XT;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (compute && j < nact) x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + (row0 + j) * ldg + d * NP + ub * 64 + u * 4));
    auto kgroup = [&](int kg) __attribute__((always_inline)) {
      const float av = hc[kg * 64 + lane];              // lane (block b, row i): h[row i][k = 16 kg + b]
#define QSTEP(jj) { const int k = kg * 16 + jj; \
        const float bw = k < RK ? wr[k < RK ? k : 0] : wlq[((long)w * QL + (k >= RK ? k - RK : 0)) * 64 + lane]; \
        acc[jj & 3] = MFMA4_BCAST(av, bw, acc[jj & 3], jj); }
      QSTEP(0) QSTEP(1) QSTEP(2) QSTEP(3) QSTEP(4) QSTEP(5) QSTEP(6) QSTEP(7) QSTEP(8) QSTEP(9) QSTEP(10) QSTEP(11) QSTEP(12) QSTEP(13) QSTEP(14) QSTEP(15)
#undef QSTEP
    };
    constexpr int KSPLIT = 4;                             // own k-groups before the partner's words are requested
    if (compute) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const int own = kg - (hv ? UB0 : 0);
        if ((kg < UB0) == (hv == 0) && own < KSPLIT) kgroup(kg);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long v = 0;
    const int frow = tid / XW, fcc = tid - frow * XW;
    const bool fetch = frow < nprev && fcc < pw;          // (4 rows x XW words <= 512 threads)
    if (fetch) v = ld_tag_first(src + tid, same_xcd);
    __builtin_amdgcn_sched_barrier(0);
    if (compute) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const int own = kg - (hv ? UB0 : 0);
        if ((kg < UB0) == (hv == 0) && own >= KSPLIT) kgroup(kg);
      }
    }
    if (fetch) hcw[(pu0 + fcc) * 4 + frow] = wait_tag(src + tid, v, a.epoch + (unsigned)step, same_xcd, diag, a.tmo_total);
    __syncthreads();                                      // the partner's half of h_{t-1} is in LDS
    if (compute) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
        if ((kg < UB0) != (hv == 0)) kgroup(kg);
      const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);      // rows 0..3 of column (unit u, gate j)
      // lane (u, j) needs gate g of row j: element j of lane (u, g)
      float gate[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float b0 = quad_bcast(sum[0], g), b1 = quad_bcast(sum[1], g), b2 = quad_bcast(sum[2], g), b3 = quad_bcast(sum[3], g);
        gate[g] = j == 0 ? b0 : (j == 1 ? b1 : (j == 2 ? b2 : b3));
      }
      const float gi = fast_sigmoid(gate[0] + x[0]);
      const float gf = fast_sigmoid(gate[1] + x[1]);
      const float gg = fast_tanh(gate[2] + x[2]);
      const float go = fast_sigmoid(gate[3] + x[3]);
      const float cn = gf * c + gi * gg;
      const float hval = go * fast_tanh(cn);
      if (j < nact) {
        if (unit < H) st_tag(xmine + (step & 1) * XT + j * XW + (unit - ub_lo * 16), hval, a.epoch + (unsigned)(step + 1), same_xcd);
        c = cn;
        hn[unit * 4 + j] = hval;
        if (unit < H) P.hout[(row0 + j) * ldh + d * H + unit] = hval;
        *reinterpret_cast<f32x4*>(P.gates + (row0 + j) * ldg + d * NP + ub * 64 + u * 4) = f32x4{gi, gf, gg, go};
        P.cell[(row0 + j) * ldc + d * HP + unit] = cn;
      }
    }
    __syncthreads();
  }
  if (compute && unit < H && s0 + j < P.n) P.cn[(long)(s0 + j) * ldh + d * H + unit] = c;
}

template <int UB>
__global__ __launch_bounds__(512) void lstm_fwd_pair_kernel(LstmArgs a) {
  // the two halves are separate instantiations: every K-range loop is static, so the LDS operand reads software-pipeline
  __shared__ __attribute__((aligned(16))) float hbuf[2][16 * UB * 16];
  __shared__ f32x4 wl[(UB + 1) / 2][UB][64];             // gate 3 ("o") fragments of every compute wave
  const PairId id = pair_id(a.nprob);
  const LstmProblem& P = a.p[id.prob];
  const int nq = quad_count(a, P);
  if (id.tile < nq) {
    if (pair_half(blockIdx.x)) lstm_fwd_quad_body<UB, 1>(a, P, id.d, id.tile, &hbuf[0][0], reinterpret_cast<float*>(&wl[0][0][0]));
    else lstm_fwd_quad_body<UB, 0>(a, P, id.d, id.tile, &hbuf[0][0], reinterpret_cast<float*>(&wl[0][0][0]));
    return;
  }
  const int s0 = 4 * nq + 16 * (id.tile - nq);
  if (pair_half(blockIdx.x)) lstm_fwd_pair_body<UB, 1>(a, P, id.d, id.tile, s0, hbuf, wl);
  else lstm_fwd_pair_body<UB, 0>(a, P, id.d, id.tile, s0, hbuf, wl);
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

// ------------------------------------------------------------------------------------------------ backward, 2 CUs per tile
// Same pairing and exchange as the forward pair kernel.  dh_{prev}[16, HP] = dgates[16, NP] . W_hh is split over K: each
// half multiplies ITS OWN gate-gradient columns (the unit blocks it owns, so the dgates tile in LDS never leaves the CU)
// with W_hh for ALL units -- one 16-unit output tile of its own per wave, plus one of the partner's.  The partner's tile
// is computed first and sent as tagged words straight from the accumulator registers; the receiving wave has the SAME
// lane layout (row = 4*(lane>>4)+e, unit = lane&15), so it adds the four words to its own partial sums in registers at
// the start of the next step: no LDS staging, no extra barrier, and the round trip hides under the own-tile MFMAs.
// W_hh fragments are resident: the own tile's K range in registers, the partner tile's half in registers, half in LDS.
// BAL (round 4): the 13 output tiles of a step (nb own + npb partner tiles per half) cannot be dealt evenly to four SIMDs as whole
// tiles -- waves w and w + 4 share a SIMD, and with waves 0..5 carrying two tiles each, wave 6 one and wave 7 none, two SIMDs did
// 4 tiles of MFMA work per step (448 instructions), the others 3 and 2: 6.0 of the step's 9.7 us.  Now the LAST own tile is split
// four ways along the reduction: waves 0-3 keep (own w, partner w), waves 4-7 take one whole tile each (the remaining own / partner
// tiles, fragments in registers) plus a QUARTER of the split tile; the four quarter sums meet in LDS and the tile's owner wave adds
// them in order at the start of the next step (where it needs them).  Every SIMD then issues 3.25 tiles = 364 MFMAs per step.
template <int UB, int hv, bool BAL>
__device__ __forceinline__ void lstm_bwd_pair_body(const LstmArgs& a, const LstmProblem& P, const int d, const int tile, const int s0,
                                                   float* dg, f32x4 (*wl)[64]) {
  constexpr int HP = UB * 16, NP = UB * 64, KGB = NP / 16, UB0 = (UB + 1) / 2, NW = 8;
  constexpr int ub_lo = hv ? UB0 : 0, nb = hv ? UB - UB0 : UB0;          // own unit blocks
  constexpr int npb = UB - nb, pb_lo = hv ? 0 : UB0;                     // the partner's unit blocks
  constexpr int KO = nb * 4, k_lo = ub_lo * 4;                           // own K range (16-column groups of p-ordered gate columns)
  constexpr int KR = BAL ? 8 : KO / 2, KL = KO - KR;                               // partner-tile fragments in registers / in LDS
  constexpr int DLD = nb * 64 + 16;                                      // row stride = 4 (mod 16) 16-byte chunks (swizzle)
  static_assert((DLD / 4) % 16 == 4, "dgates tile stride must keep the ds_read_b128 swizzle conflict-free");
  const int H = a.H;
  if (s0 >= P.n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, kk = lane >> 4;
  const int tmax = P.slen[s0];
  if (tmax == 1) {
    // one-token sequences (see the forward body): the only step is the last forward step of every row, there is no previous
    // state to send a gradient to -- element-wise gate gradients, no weights, no partner
    if (w < nb) {
      const int ub = ub_lo + w, unit = ub * 16 + r;
      const int nact = min(16, P.bs[0] - s0);
      const long row0 = (long)P.off[0] + s0;
      const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        if (row < nact) {
          const long grow = row0 + row;
          float* gp = P.gates + grow * ldg + d * NP + ub * 64 + r * 4;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp);
          const float gi = g4[0], gg = g4[2], go = g4[3];
          const float ct = P.cell[grow * ldc + d * HP + unit];
          const float dh = unit < H ? P.dh[grow * ldh + d * H + unit] : 0.f;
          float dc = (P.dcn && unit < H) ? P.dcn[(long)(s0 + row) * ldh + d * H + unit] : 0.f;
          const float tc = fast_tanh(ct);
          f32x4 dgv;
          dgv[3] = dh * tc * go * (1.f - go);
          dc += dh * go * (1.f - tc * tc);
          dgv[0] = dc * gg * gi * (1.f - gi);
          dgv[1] = 0.f;                                  // c_0 = 0
          dgv[2] = dc * gi * (1.f - gg * gg);
          *reinterpret_cast<f32x4*>(gp) = dgv;
        }
      }
    }
    return;
  }
  const int ntiles = sync_tiles(P.n);
  constexpr int XT = UB0 * 4 * 64, XS = 2 * XT + 8;                      // [step parity][tile][e][lane] tagged words (+ handshake line)
  unsigned long long* xmine = reinterpret_cast<unsigned long long*>(P.sync) + ((long)(d * ntiles + tile) * 2 + hv) * XS;
  const unsigned long long* xtheirs = reinterpret_cast<const unsigned long long*>(P.sync) + ((long)(d * ntiles + tile) * 2 + (hv ^ 1)) * XS;
  unsigned* diag = P.sync + (long)2 * ntiles * 2 * XS * 2;
  if (tid == 0) {
    const unsigned mine = xcc_id();
    st_tag(xmine + 2 * XT, __uint_as_float(mine), a.epoch + 0x3ffu, false);
    const float theirs = wait_tag(xtheirs + 2 * XT, ld_tag(xtheirs + 2 * XT, false), a.epoch + 0x3ffu, false, diag, a.tmo_total);
    dg[0] = (__float_as_uint(theirs) == mine && !(a.dbg & 64)) ? 1.f : 0.f;
  }
  __syncthreads();
  const bool same_xcd = __builtin_amdgcn_readfirstlane(dg[0] != 0.f);
  __syncthreads();

  unsigned long long* tbuf = reinterpret_cast<unsigned long long*>(diag + SYNC_PAD);
  const bool stamp = (a.dbg & 32) && blockIdx.x == 0 && tid == 0;
#define STAMP(slot) do { if (stamp && step < 128) tbuf[step * 16 + (slot)] = wall_clock64(); } while (0)
  const bool own = w < nb;                               // this wave owns unit block ub_lo + w (gate gradients; legacy: + its output tile)
  constexpr int SPL = nb - 1;                            // BAL: the own tile that is split four ways; wave SPL is its owner
  constexpr int KQ4 = KO / 4;                            // k-groups per quarter
  static_assert(!BAL || (KO % 4 == 0 && nb >= 6 && npb >= 6 && nb <= 7 && npb <= 7 && KQ4 <= KR), "balanced split is laid out for 13 unit blocks");
  const bool hi = BAL && w >= 4;                         // BAL, waves 4-7: one whole tile (registers) + quarter w - 4 of the split tile
  const bool t1_own = BAL ? (w < 4 || w < SPL) : own;    // the tile whose fragments sit in `wo` is an own tile (result stays in dhr)
  const int t1_tile = (BAL && w >= SPL) ? 4 + (w - SPL) : w;      // ... its index among the own / the partner's tiles
  const bool t1_on = BAL ? true : own;
  const bool par = BAL ? (w < 4) : (w < npb);            // (wp / wl fragments) the partner's output tile pb_lo + w
  if (!BAL && !own && !par) {                            // nothing to do: keep the barrier count
    for (int step = 0; step < tmax; ++step) { __syncthreads(); __syncthreads(); }
    return;
  }
  const int ub = ub_lo + (own ? w : 0);
  const int unit = ub * 16 + r;
  const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
  int mylen[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int s = s0 + kk * 4 + e;
    mylen[e] = (s < P.n) ? P.slen[s] : 0;
  }
  // resident weights: wb[d][n-tile][kg][lane] (lstm_pack_kernel)
  f32x4 wo[KO], wp[KR];
  {
    const f32x4* wb = reinterpret_cast<const f32x4*>(P.wfrag) + (long)d * UB * KGB * 64 + lane;
    const int t1_blk = (t1_own ? ub_lo : pb_lo) + t1_tile;       // unit block (n-tile of W_hh^T) of the whole tile kept in `wo`
#pragma unroll
    for (int k = 0; k < KO; ++k) wo[k] = t1_on ? wb[((long)t1_blk * KGB + k_lo + k) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    const int pb = pb_lo + (par ? w : 0);
#pragma unroll
    for (int k = 0; k < KR; ++k) wp[k] = par ? wb[((long)pb * KGB + k_lo + k) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    if (par) {
#pragma unroll
      for (int k = 0; k < KL; ++k) wl[w * KL + k][lane] = wb[((long)pb * KGB + k_lo + KR + k) * 64];
    }
    if (hi) {                                                // quarter w - 4 of the split own tile: KQ4 fragments, in wp's registers
#pragma unroll
      for (int k = 0; k < KR; ++k)
        if (k < KQ4) wp[k] = wb[((long)(ub_lo + SPL) * KGB + k_lo + (w - 4) * KQ4 + k) * 64];
    }
  }
  f32x4 (*qs)[64] = wl + 4 * KL;                             // BAL: the four quarter sums of the split tile (rows of wl nobody else uses)
  float dhr[4] = {0.f, 0.f, 0.f, 0.f}, dcr[4] = {0.f, 0.f, 0.f, 0.f};
  // software-pipelined inputs of the gate-gradient phase (loaded under the MFMA phase of the previous step)
  f32x4 in_g[4];
  float in_ct[4], in_cp[4], in_dh[4];
  auto load_inputs = [&](int step) __attribute__((always_inline)) {
    const int t = d ? step : (tmax - 1 - step);
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    const int tp = d ? t + 1 : t - 1;
    const long prow0 = (tp >= 0 && tp < P.L) ? (long)P.off[tp] + s0 : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = kk * 4 + e;
      in_g[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      in_ct[e] = 0.f; in_cp[e] = 0.f; in_dh[e] = 0.f;
      if (own && row < nact) {
        const long grow = row0 + row;
        in_g[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + grow * ldg + d * NP + ub * 64 + r * 4));
        in_ct[e] = __builtin_nontemporal_load(P.cell + grow * ldc + d * HP + unit);
        const bool has_prev = d ? (t + 1 < mylen[e]) : (t > 0);
        if (has_prev) in_cp[e] = __builtin_nontemporal_load(P.cell + (prow0 + row) * ldc + d * HP + unit);
        if (unit < H) in_dh[e] = __builtin_nontemporal_load(P.dh + grow * ldh + d * H + unit);
      }
    }
  };
  load_inputs(0);

  for (int step = 0; step < tmax; ++step) {
    const int t = d ? step : (tmax - 1 - step);
    const int nact = min(16, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    // ---- (1) gate gradients of the own unit block
    STAMP(4);
    if (own) {
      if (BAL && w == SPL && step > 0) {
        // this wave's own output tile of the previous step was computed in four K-quarters by waves 4-7: add them in order
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = qs[q][lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) dhr[e] = q ? dhr[e] + v[e] : v[e];
        }
      }
      // the partner's contribution to dh of these units (its own-K partial sums of the previous step)
      if (step > 0) {
        const unsigned long long* src = xtheirs + ((step + 1) & 1) * XT + (w * 4) * 64 + lane;
        unsigned long long v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ld_tag_first(src + e * 64, same_xcd);
#pragma unroll
        for (int e = 0; e < 4; ++e) dhr[e] += wait_tag(src + e * 64, v[e], a.epoch + (unsigned)step, same_xcd, diag, a.tmo_total);
      }
      if (stamp && step < 128) tbuf[step * 16 + 5] = wall_clock64() + (unsigned long long)(dhr[0] == 123.456f);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = kk * 4 + e;
        f32x4 dgv = {0.f, 0.f, 0.f, 0.f};
        if (row < nact) {
          const long grow = row0 + row;
          const float gi = in_g[e][0], gf = in_g[e][1], gg = in_g[e][2], go = in_g[e][3];
          const float ct = in_ct[e], cp = in_cp[e];
          const float dh = dhr[e] + in_dh[e];
          float dc = dcr[e];
          const bool last_fwd_step = d ? (t == 0) : (t == mylen[e] - 1);
          if (last_fwd_step && P.dcn && unit < H) dc += P.dcn[(long)(s0 + row) * ldh + d * H + unit];
          const float tc = fast_tanh(ct);
          dgv[3] = dh * tc * go * (1.f - go);
          dc += dh * go * (1.f - tc * tc);
          dgv[0] = dc * gg * gi * (1.f - gi);
          dgv[1] = dc * cp * gf * (1.f - gf);
          dgv[2] = dc * gi * (1.f - gg * gg);
          dcr[e] = dc * gf;
          *reinterpret_cast<f32x4*>(P.gates + grow * ldg + d * NP + ub * 64 + r * 4) = dgv;
        }
        *reinterpret_cast<f32x4*>(&dg[lds_off(row, w * 64 + r * 4, DLD)]) = dgv;
      }
    }
    STAMP(6);
    __syncthreads();
    STAMP(7);
    if (step + 1 < tmax) load_inputs(step + 1);
    // ---- (2) the partner's output tile over the own K range: compute, send
    if (par) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KO; ++k) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(&dg[r * DLD + k * 16 + 4 * (kk ^ swz16(r))]);
        const f32x4 b = k < KR ? wp[k < KR ? k : 0] : wl[w * KL + (k >= KR ? k - KR : 0)][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], b[i], acc, 0, 0, 0);
      }
      unsigned long long* dst = xmine + (step & 1) * XT + (w * 4) * 64 + lane;
#pragma unroll
      for (int e = 0; e < 4; ++e) st_tag(dst + e * 64, acc[e], a.epoch + (unsigned)(step + 1), same_xcd);
      STAMP(8);
    }
    // ---- (3) the whole tile whose fragments are resident in registers: an own tile stays in registers (the same lane needs it next
    //      step), a partner tile (BAL, waves >= SPL) is sent like the ones above
    if (t1_on) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KO; ++k) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(&dg[r * DLD + k * 16 + 4 * (kk ^ swz16(r))]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wo[k][i], acc, 0, 0, 0);
      }
      if (t1_own) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dhr[e] = acc[e];
      } else {
        unsigned long long* dst = xmine + (step & 1) * XT + (t1_tile * 4) * 64 + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) st_tag(dst + e * 64, acc[e], a.epoch + (unsigned)(step + 1), same_xcd);
      }
      if (stamp && step < 128) tbuf[step * 16 + 9] = wall_clock64() + (unsigned long long)(dhr[0] == 123.456f);
    }
    // ---- (4) BAL, waves 4-7: quarter w - 4 of the split own tile
    if (hi) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        if (k < KQ4) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(&dg[r * DLD + ((w - 4) * KQ4 + k) * 16 + 4 * (kk ^ swz16(r))]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wp[k][i], acc, 0, 0, 0);
        }
      }
      qs[w - 4][lane] = acc;
    }
    __syncthreads();
  }
#undef STAMP
}

// ---- backward, quad tile (4 sequences) on a CU pair: the 16-row body's split (own gate-gradient columns as the K range, the
// partner's output tile first and sent from registers, the own tile second), on v_mfma_f32_4x4x1.  An output tile is 16 units;
// the 16 blocks of an instruction are 4 unit-quads x 4 QUARTERS of the K range (every block multiplies its own A: rows of dgates
// at the k of its quarter), so one instruction still does 64 columns of work; the four quarter sums meet through two
// cross-lane adds.  Cell-gradient lanes: (row = lane >> 4, unit = lane & 15).
template <int UB, int hv>
__device__ __forceinline__ void lstm_bwd_quad_body(const LstmArgs& a, const LstmProblem& P, const int d, const int q, float* dgq, float* wlq) {
  constexpr int HP = UB * 16, NP = UB * 64, KGB = NP / 16, UB0 = (UB + 1) / 2;
  constexpr int ub_lo = hv ? UB0 : 0, nb = hv ? UB - UB0 : UB0;
  constexpr int npb = UB - nb, pb_lo = hv ? 0 : UB0;
  constexpr int KO = nb * 64, KQ = KO / 4, p_lo = ub_lo * 64;            // own gate-gradient columns, per quarter
  constexpr int KR = KQ / 2, KL = KQ - KR;                               // partner-tile weights in registers / in LDS
  constexpr int DLD = KO + 4;
  const int H = a.H;
  const int s0 = q * 4;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = lane >> 4, u = lane & 15;                              // cell-gradient role; as MFMA column: quarter `row`, unit u
  const int tmax = P.slen[s0];
  __builtin_amdgcn_s_setprio(3);
  const int ntl = sync_tiles(P.n);
  constexpr int XT = UB0 * 4 * 64, XS = 2 * XT + 8;
  unsigned long long* xmine = reinterpret_cast<unsigned long long*>(P.sync) + ((long)(d * ntl + q) * 2 + hv) * XS;
  const unsigned long long* xtheirs = reinterpret_cast<const unsigned long long*>(P.sync) + ((long)(d * ntl + q) * 2 + (hv ^ 1)) * XS;
  unsigned* diag = P.sync + (long)2 * ntl * 2 * XS * 2;
  if (tid == 0) {
    const unsigned mine = xcc_id();
    st_tag(xmine + 2 * XT, __uint_as_float(mine), a.epoch + 0x3ffu, false);
    const float theirs = wait_tag(xtheirs + 2 * XT, ld_tag(xtheirs + 2 * XT, false), a.epoch + 0x3ffu, false, diag, a.tmo_total);
    dgq[0] = (__float_as_uint(theirs) == mine && !(a.dbg & 64)) ? 1.f : 0.f;
  }
  __syncthreads();
  const bool same_xcd = __builtin_amdgcn_readfirstlane(dgq[0] != 0.f);
  __syncthreads();
  const bool own = w < nb;
  const bool par = w < npb;
  if (!own && !par) {
    for (int step = 0; step < tmax; ++step) { __syncthreads(); __syncthreads(); }
    return;
  }
  const int ub = ub_lo + (own ? w : 0);
  const int unit = ub * 16 + u;
  const int ldg = 2 * NP, ldc = 2 * HP, ldh = 2 * H;
  const int mylen = (s0 + row < P.n) ? P.slen[s0 + row] : 0;
  // resident weights from wb[d][n-tile][kg][lane' = unit + 16 * ((p >> 2) & 3)][p & 3] (the 16-row kernels' fragment layout):
  // lane (quarter, u) keeps W_hh_p[p = p_lo + quarter * KQ + k][unit] for k < KQ
  float wo[KQ], wp[KR];
  {
    const float* wb = P.wfrag + (long)d * UB * KGB * 256 + u * 4;
    const int pq = p_lo + row * KQ;
    const int pb = pb_lo + (par ? w : 0);
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      const int pp = pq + k;
      wo[k] = own ? wb[((long)ub * KGB + (pp >> 4)) * 256 + ((pp >> 2) & 3) * 64 + (pp & 3)] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int pp = pq + k;
      wp[k] = par ? wb[((long)pb * KGB + (pp >> 4)) * 256 + ((pp >> 2) & 3) * 64 + (pp & 3)] : 0.f;
    }
    if (par) {
#pragma unroll 4
      for (int k = KR; k < KQ; ++k) {
        const int pp = pq + k;
        wlq[((long)w * KL + (k - KR)) * 64 + lane] = wb[((long)pb * KGB + (pp >> 4)) * 256 + ((pp >> 2) & 3) * 64 + (pp & 3)];
      }
    }
  }
  float dhr = 0.f, dcr = 0.f;
  f32x4 in_g;
  float in_ct, in_cp, in_dh;
  auto load_inputs = [&](int step) __attribute__((always_inline)) {
    const int t = d ? step : (tmax - 1 - step);
    const int nact = min(4, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    const int tp = d ? t + 1 : t - 1;
    const long prow0 = (tp >= 0 && tp < P.L) ? (long)P.off[tp] + s0 : 0;
    in_g = f32x4{0.f, 0.f, 0.f, 0.f};
    in_ct = 0.f; in_cp = 0.f; in_dh = 0.f;
    if (own && row < nact) {
      const long grow = row0 + row;
      in_g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.gates + grow * ldg + d * NP + ub * 64 + u * 4));
      in_ct = __builtin_nontemporal_load(P.cell + grow * ldc + d * HP + unit);
      const bool has_prev = d ? (t + 1 < mylen) : (t > 0);
      if (has_prev) in_cp = __builtin_nontemporal_load(P.cell + (prow0 + row) * ldc + d * HP + unit);
      if (unit < H) in_dh = __builtin_nontemporal_load(P.dh + grow * ldh + d * H + unit);
    }
  };
  load_inputs(0);
  // one output tile (16 units) over the own K range: 4 quarters side by side, then the quarter sums are added across lanes
  auto tile_mfma = [&](auto weight) __attribute__((always_inline)) -> float {
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < KQ / 4; ++kq) {
      // A of block b = lane >> 2 (quarter b >> 2 = lane >> 4), row lane & 3: dgates[row][quarter * KQ + 4 kq ..]
      const f32x4 af = *reinterpret_cast<const f32x4*>(&dgq[(lane & 3) * DLD + (lane >> 4) * KQ + kq * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = MFMA4_OWN(af[i], weight(kq * 4 + i), acc[i]);
    }
    f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sum[e] += __shfl_xor(sum[e], 16, 64);
      sum[e] += __shfl_xor(sum[e], 32, 64);
    }
    return row == 0 ? sum[0] : (row == 1 ? sum[1] : (row == 2 ? sum[2] : sum[3]));      // lane (row, u): dh_prev[row][tile unit u]
  };

  for (int step = 0; step < tmax; ++step) {
    const int t = d ? step : (tmax - 1 - step);
    const int nact = min(4, P.bs[t] - s0);
    const long row0 = (long)P.off[t] + s0;
    if (own) {
      if (step > 0) {
        const unsigned long long* src = xtheirs + ((step + 1) & 1) * XT + w * 64 + lane;
        dhr += wait_tag(src, ld_tag_first(src, same_xcd), a.epoch + (unsigned)step, same_xcd, diag, a.tmo_total);
      }
      f32x4 dgv = {0.f, 0.f, 0.f, 0.f};
      if (row < nact) {
        const long grow = row0 + row;
        const float gi = in_g[0], gf = in_g[1], gg = in_g[2], go = in_g[3];
        const float ct = in_ct, cp = in_cp;
        const float dh = dhr + in_dh;
        float dc = dcr;
        const bool last_fwd_step = d ? (t == 0) : (t == mylen - 1);
        if (last_fwd_step && P.dcn && unit < H) dc += P.dcn[(long)(s0 + row) * ldh + d * H + unit];
        const float tc = fast_tanh(ct);
        dgv[3] = dh * tc * go * (1.f - go);
        dc += dh * go * (1.f - tc * tc);
        dgv[0] = dc * gg * gi * (1.f - gi);
        dgv[1] = dc * cp * gf * (1.f - gf);
        dgv[2] = dc * gi * (1.f - gg * gg);
        dcr = dc * gf;
        *reinterpret_cast<f32x4*>(P.gates + grow * ldg + d * NP + ub * 64 + u * 4) = dgv;
      }
      *reinterpret_cast<f32x4*>(&dgq[row * DLD + w * 64 + u * 4]) = dgv;
    }
    __syncthreads();
    if (step + 1 < tmax) load_inputs(step + 1);
    if (par) {
      const float val = tile_mfma([&](int k) __attribute__((always_inline)) { return k < KR ? wp[k < KR ? k : 0] : wlq[((long)w * KL + (k >= KR ? k - KR : 0)) * 64 + lane]; });
      st_tag(xmine + (step & 1) * XT + w * 64 + lane, val, a.epoch + (unsigned)(step + 1), same_xcd);
    }
    if (own) dhr = tile_mfma([&](int k) __attribute__((always_inline)) { return wo[k]; });
    __syncthreads();
  }
}

template <int UB, bool BAL>
__global__ __launch_bounds__(512) void lstm_bwd_pair_kernel(LstmArgs a) {
  constexpr int UB0 = (UB + 1) / 2;
  __shared__ __attribute__((aligned(16))) float dg[16 * (UB0 * 64 + 16)];
  __shared__ f32x4 wl[UB0 * (UB0 * 4 - UB0 * 2)][64];    // partner-tile fragments kept in LDS: [wave][KL]
  const PairId id = pair_id(a.nprob);
  const LstmProblem& P = a.p[id.prob];
  const int nq = quad_count(a, P);
  if (id.tile < nq) {
    if (pair_half(blockIdx.x)) lstm_bwd_quad_body<UB, 1>(a, P, id.d, id.tile, dg, reinterpret_cast<float*>(&wl[0][0]));
    else lstm_bwd_quad_body<UB, 0>(a, P, id.d, id.tile, dg, reinterpret_cast<float*>(&wl[0][0]));
    return;
  }
  const int s0 = 4 * nq + 16 * (id.tile - nq);
  if (pair_half(blockIdx.x)) lstm_bwd_pair_body<UB, 1, BAL>(a, P, id.d, id.tile, s0, dg, wl);
  else lstm_bwd_pair_body<UB, 0, BAL>(a, P, id.d, id.tile, s0, dg, wl);
}

// ------------------------------------------------------------------------------------------------ weight (un)packing
__global__ void lstm_pack_kernel(const float* __restrict__ w_ih_f, const float* __restrict__ w_hh_f,
                                 const float* __restrict__ b_ih_f, const float* __restrict__ b_hh_f,
                                 const float* __restrict__ w_ih_r, const float* __restrict__ w_hh_r,
                                 const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r, int H, int E, int UB,
                                 float* __restrict__ w_ihp, float* __restrict__ b_p, float* __restrict__ wf,
                                 float* __restrict__ wb, float* __restrict__ w_ihp_t) {
  const int NP = UB * 64, KG = UB, KGB = NP / 16;
  const long n_ihp = (long)2 * NP * E, n_b = 2 * NP, n_wf = (long)2 * UB * 4 * KG * 256, n_wb = (long)2 * UB * KGB * 256;
  const long total = n_ihp + n_b + n_wf + n_wb;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long i = idx;
    if (i < n_ihp) {                       // w_ihp[d*NP + p][e]
      const int e = i % E; const int dp = i / E; const int d = dp / NP, p = dp % NP;
      const int ub = p / 64, u = (p % 64) / 4, g = p % 4, unit = ub * 16 + u;
      const float* src = d ? w_ih_r : w_ih_f;
      const float v = (unit < H) ? src[(long)(g * H + unit) * E + e] : 0.f;
      w_ihp[i] = v;
      if (w_ihp_t) w_ihp_t[(long)e * (2 * NP) + dp] = v;      // [E, 2 NP]: the K-contiguous B operand of the dX GEMM (NT form)
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
__global__ void lstm_unpack_kernel(float* __restrict__ dw_ihp, float* __restrict__ db_p,
                                   float* __restrict__ dw_hhp, int H, int E, int UB, float* __restrict__ dw_ih_f,
                                   float* __restrict__ dw_hh_f, float* __restrict__ db_ih_f, float* __restrict__ db_hh_f,
                                   float* __restrict__ dw_ih_r, float* __restrict__ dw_hh_r, float* __restrict__ db_ih_r,
                                   float* __restrict__ db_hh_r, int accumulate, int zero_src) {
  // accumulate: atomic += into the running parameter gradients (several token streams / HIP streams add concurrently)
  // zero_src: the packed buffers are a PERSISTENT workspace that the next step's split-K GEMMs add into again: every element
  // is zeroed by the thread that consumed it (the padded ones, which nothing reads, by the loop at the end)
  auto put = [&](float* p, float v) __attribute__((always_inline)) { if (accumulate) atomicAdd(p, v); else *p = v; };
  const int NP = UB * 64;
  const long n_ih = (long)2 * 4 * H * E, n_hh = (long)2 * 4 * H * H, n_b = 2 * 4 * H;
  const long total = n_ih + n_hh + n_b;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long i = idx;
    if (i < n_ih) {
      const int e = i % E; long q = i / E; const int row = q % (4 * H); const int d = q / (4 * H);
      const int g = row / H, unit = row % H, p = (unit / 16) * 64 + (unit % 16) * 4 + g;
      float* src = &dw_ihp[(long)(d * NP + p) * E + e];
      put(&(d ? dw_ih_r : dw_ih_f)[(long)row * E + e], *src);
      if (zero_src) *src = 0.f;
      continue;
    }
    i -= n_ih;
    if (i < n_hh) {
      const int k = i % H; long q = i / H; const int row = q % (4 * H); const int d = q / (4 * H);
      const int g = row / H, unit = row % H, p = (unit / 16) * 64 + (unit % 16) * 4 + g;
      float* src = &dw_hhp[((long)d * NP + p) * H + k];
      put(&(d ? dw_hh_r : dw_hh_f)[(long)row * H + k], *src);
      if (zero_src) *src = 0.f;
      continue;
    }
    i -= n_hh;
    {
      const int row = i % (4 * H), d = i / (4 * H);
      const int g = row / H, unit = row % H, p = (unit / 16) * 64 + (unit % 16) * 4 + g;
      const float v = db_p[d * NP + p];
      put(&(d ? db_ih_r : db_ih_f)[row], v);
      put(&(d ? db_hh_r : db_hh_f)[row], v);
      if (zero_src) db_p[d * NP + p] = 0.f;
    }
  }
  if (zero_src && (UB * 16 != H)) {          // padded gate columns (unit >= H): never read above, but the GEMMs add into them
    const long n1 = (long)2 * NP * E, n2 = (long)2 * NP * H, n3 = 2 * NP;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n1 + n2 + n3; idx += (long)gridDim.x * blockDim.x) {
      long i = idx;
      float* base = dw_ihp; int width = E;
      if (i >= n1) { i -= n1; base = dw_hhp; width = H; if (i >= n2) { i -= n2; base = db_p; width = 1; } }
      const int p = (int)((i / width) % NP);
      const int unit = (p / 64) * 16 + ((p % 64) / 4);
      if (unit >= H) base[i] = 0.f;
    }
  }
}

template <int UB>
int launch_pair(const LstmArgs& a, bool backward, int max_tiles, hipStream_t s) {
  const int tiles = max_tiles + (a.quad_T > 0 ? MAXQ : 0);             // quad tiles take the first ids; whatever is not needed exits
  dim3 grid(((tiles + 7) / 8) * 16 * 2 * a.nprob), block(512);         // groups of 8 tiles x 2 halves, see pair_id()
  static const bool bal = [] { const char* e = getenv("NNR_LSTM_BWD_BAL"); return !(e && atoi(e) == 0); }();      // A/B: 0 = the legacy whole-tile split
  if (backward) {
    if constexpr (UB == 13) {
      if (bal) hipLaunchKernelGGL((lstm_bwd_pair_kernel<UB, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((lstm_bwd_pair_kernel<UB, false>), grid, block, 0, s, a);
    } else {
      hipLaunchKernelGGL((lstm_bwd_pair_kernel<UB, false>), grid, block, 0, s, a);
    }
  } else hipLaunchKernelGGL((lstm_fwd_pair_kernel<UB>), grid, block, 0, s, a);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
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
  return (ub >= 1 && ub <= 16) ? NNR_OK : NNR_ERR_UNSUPPORTED;      // any hidden size up to 256 (padded to a multiple of 16)
}

static unsigned* g_tmo_total = nullptr;     // caller-owned persistent time-out counter (device memory), see nnr_lstm_set_timeout_counter

extern "C" int nnr_lstm_set_timeout_counter(unsigned* dev_counter) {
  g_tmo_total = dev_counter;
  return NNR_OK;
}

extern "C" size_t nnr_lstm_sync_diag_offset(int n) {
  // byte offset of the diagnostics block (word 0 = exchange time-outs of the LAST launch on this workspace) inside the
  // workspace of nnr_lstm_sync_bytes(n): it sits behind the exchange slots, in front of the 128 x 16 stamp words
  return nnr_lstm_sync_bytes(n) - SYNC_PAD * sizeof(unsigned) - 128 * 16 * 8;
}

extern "C" size_t nnr_lstm_sync_bytes(int n) {
  const size_t ntiles = (size_t)(n + 15) / 16;
  // [2 directions][ntiles][2 halves][2 step parities][16 rows][7 * 16 units] tagged 8-byte words + 64 bytes of diagnostics
  return 2 * (ntiles + MAXQ) * 2 * (2 * 16 * 112 + 8) * 8 + SYNC_PAD * sizeof(unsigned) + 128 * 16 * 8;      // (+ MAXQ quad-tile slots)
}

extern "C" int nnr_lstm_pack_weights(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                                     const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                     int H, int E, float* w_ihp, float* b_p, float* wf, float* wb, float* w_ihp_t, hipStream_t stream) {
  int UB;
  if (nnr_lstm_dims(H, &UB, nullptr, nullptr) != NNR_OK) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(1024), dim3(256), 0, stream, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r,
                     b_ih_r, b_hh_r, H, E, UB, w_ihp, b_p, wf, wb, w_ihp_t);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_lstm_unpack_grads(float* dw_ihp, float* db_p, float* dw_hhp, int H, int E,
                                     float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f, float* dw_ih_r,
                                     float* dw_hh_r, float* db_ih_r, float* db_hh_r, int accumulate, int zero_src, hipStream_t stream) {
  int UB;
  if (nnr_lstm_dims(H, &UB, nullptr, nullptr) != NNR_OK) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lstm_unpack_kernel, dim3(1024), dim3(256), 0, stream, dw_ihp, db_p, dw_hhp, H, E, UB, dw_ih_f,
                     dw_hh_f, db_ih_f, db_hh_f, dw_ih_r, dw_hh_r, db_ih_r, db_hh_r, accumulate, zero_src);
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
  a.tmo_total = g_tmo_total;
  int max_tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const nnr_lstm_problem& q = probs[i];
    LstmProblem& p = a.p[i];
    p.bs = q.bs; p.off = q.off; p.slen = q.slen; p.prev_f = q.prev_f; p.prev_r = q.prev_r;
    p.n = q.n; p.L = q.L;
    p.gates = q.gates; p.cell = q.cell; p.hout = q.hout; p.cn = q.cn;
    p.wfrag = backward ? q.wb : q.wf;
    p.dh = q.dh; p.dcn = q.dcn; p.sync = q.sync;
    if (!p.bs || !p.off || !p.slen || !p.gates || !p.cell || !p.wfrag || p.n <= 0) return NNR_ERR_ARG;
    if (backward ? (!p.dh || !p.prev_f || !p.prev_r) : (!p.hout || !p.cn)) return NNR_ERR_ARG;
    max_tiles = max(max_tiles, (p.n + 15) / 16);
  }
  for (int i = nprob; i < 4; ++i) a.p[i] = a.p[0];
  {
    // 4-row tiles for sequences longer than quad_T steps (0: off).  They buy latency with throughput (3.3 us per step of 4 rows vs
    // 8.9 us per step of 16), so they pay while the launch is bound by its longest chain, not once the pair slots are oversubscribed
    // several times: same-box A/B of the whole step (ms, T = 0 / 16 / 32 / 64): batch 8 (440 sequences per stream) 4.85 / 3.53 /
    // 3.76 / 3.86; batch 16: 5.83 / 4.81 / 4.89 / 4.90; batch 32: 7.72 / 7.26 / 7.24 / 7.10; batch 64 (3 520): 11.66 / - / 11.77 / 11.75.
    // Only the few sequences of more than 96 steps at batch 64 and beyond (a dozen quad tiles shorten the 128-step chain the forward launch
    // is bound by: 1.11 -> 0.91 ms; T = 0 / 64 / 96 / 112: 11.58 / 11.63 / 11.46 / 11.47 ms per step once one-token tiles are element-wise).
    int maxn = 0;
    for (int i = 0; i < nprob; ++i) maxn = max(maxn, a.p[i].n);
    const char* e = getenv("NNR_LSTM_QUAD_T");
    a.quad_T = e ? atoi(e) : (maxn <= 1024 ? 16 : (maxn <= 2048 ? 64 : 96));
    if (backward) { const char* eb = getenv("NNR_LSTM_QUAD_T_BWD"); if (eb) a.quad_T = atoi(eb); }      // (A/B: own threshold for the backward launches)
  }
  // 2-CU weights-stationary recurrence when the caller provides the exchange workspace
  bool pair = UB == 13 && (H % 2 == 0);
  for (int i = 0; i < nprob; ++i) pair = pair && a.p[i].sync != nullptr;
  { const char* e = getenv("NNR_LSTM_PAIR"); if (e && atoi(e) == 0) pair = false; }
  if (pair) {
    // The launch epoch below is HOST state copied into the kernel arguments: a hipGraph that captured this launch would replay it
    // with the captured epoch, and a reader could then accept the previous replay's exchange words before the partner rewrites them.
    // While the stream is capturing, take the one-CU kernel (no exchange).  (The launch tape of tape.hip is not a capture: it calls
    // this entry point again on every replay.)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) pair = false;
  }
  if (pair) {
    // Exchange words carry (launch epoch, step) tags, so words of earlier launches never match and the 25-32 MB workspace is NOT
    // cleared per launch (4 x 31 us of fill kernels per step, each in front of a recurrence launch): only its diagnostics block is.
    // Contract: the caller zero-fills a workspace ONCE, before its first launch.
    static unsigned launch_counter = 0;
    unsigned ep = (++launch_counter) & 0x3fffffu;
    if (ep == 0) {
      // the 22-bit epoch wrapped (4 M launches): words that were never rewritten since epoch e could match launch e again.  Start
      // the next lap from clean workspaces (zero = "no word of any epoch": epoch 0 is never issued).
      for (int i = 0; i < nprob; ++i)
        if (hipMemsetAsync(a.p[i].sync, 0, nnr_lstm_sync_bytes(a.p[i].n), stream) != hipSuccess) return NNR_ERR_LAUNCH;
      ep = (++launch_counter) & 0x3fffffu;
    }
    a.epoch = ep << 10;
    for (int i = 0; i < nprob; ++i)
      if (hipMemsetAsync(reinterpret_cast<char*>(a.p[i].sync) + nnr_lstm_sync_diag_offset(a.p[i].n), 0, SYNC_PAD * sizeof(unsigned), stream) != hipSuccess)
        return NNR_ERR_LAUNCH;
    return launch_pair<13>(a, backward, max_tiles, stream);
  }
  switch (UB) {
    // one-CU kernel for every unit-block count (hidden_dim <= 256, config.py:62); H = 200 normally takes the CU-pair kernel above
    case 1: return launch_rec<1>(a, backward, max_tiles, stream);
    case 2: return launch_rec<2>(a, backward, max_tiles, stream);
    case 3: return launch_rec<3>(a, backward, max_tiles, stream);
    case 4: return launch_rec<4>(a, backward, max_tiles, stream);
    case 5: return launch_rec<5>(a, backward, max_tiles, stream);
    case 6: return launch_rec<6>(a, backward, max_tiles, stream);
    case 7: return launch_rec<7>(a, backward, max_tiles, stream);
    case 8: return launch_rec<8>(a, backward, max_tiles, stream);
    case 9: return launch_rec<9>(a, backward, max_tiles, stream);
    case 10: return launch_rec<10>(a, backward, max_tiles, stream);
    case 11: return launch_rec<11>(a, backward, max_tiles, stream);
    case 12: return launch_rec<12>(a, backward, max_tiles, stream);
    case 13: return launch_rec<13>(a, backward, max_tiles, stream);
    case 14: return launch_rec<14>(a, backward, max_tiles, stream);
    case 15: return launch_rec<15>(a, backward, max_tiles, stream);
    case 16: return launch_rec<16>(a, backward, max_tiles, stream);
  }
  return NNR_ERR_UNSUPPORTED;
}

extern "C" int nnr_lstm_fwd(const nnr_lstm_problem* probs, int nprob, int H, hipStream_t stream) {
  return lstm_run(probs, nprob, H, false, stream);
}
extern "C" int nnr_lstm_bwd(const nnr_lstm_problem* probs, int nprob, int H, hipStream_t stream) {
  return lstm_run(probs, nprob, H, true, stream);
}
