// Filtered 1-vs-all ranking: the evaluation half of the hot path.
//
// Replaces (reference repo) KGEkeras/utils.py:199-264 `validate`: per test triple the reference builds the list of
// all entities minus the target minus the filtered ones (gen_tail_data / gen_head_data, :165-196), scores it with
// model.predict (:211,231) and takes rankdata(res,'max') (:212,232), i.e.
//     real_rank = 1 + #{kept candidates c : score_c > score_target}                    (strict; ties optimistic)
//
// Here, for DistMult and ComplEx the score of (query q, candidate c) is the dot product of a per-query vector
// Qv[q] with the candidate's table row (DistMult tail: s*p, head: p*o; ComplEx: the two products derived from
// models.py:218-222), so the whole sweep is Qv . E^T on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-for-bit a k-ordered fmaf chain) with the count-greater test fused into the epilogue: the score matrix never
// exists in memory.  The target score and the scores of the (few) filtered candidates are recomputed by scalar
// kernels that run the SAME fmaf chain in the SAME k order, so "score_c > score_target" is decided on bit-identical
// numbers everywhere and a candidate can be un-counted exactly.
//
//   rank_build_q_kernel   : Qv[q][0..Kpad) (zero padded)                                    one sub-group per query
//   rank_target_kernel    : tgt[q] = chain(Qv[q], E[target_q]); counts[q] = 0               one thread per query
//   rank_sweep_kernel     : counts[q] += #{c in [0,rows) : chain(Qv[q], Ec[c]) > tgt[q]}    MFMA, 128x128 block tiles
//   rank_filter_kernel    : counts[q] -= [chain(Qv[q], E[c]) > tgt[q]] for filtered (q,c)   one thread per CSR entry
//
// k order of the chain (what one 32x32x2 MFMA stream over an LDS tile read with ds_read_b128 produces): inside every
// aligned block of 8 k-values the order is 0,4,1,5,2,6,3,7 (lane half 0 holds k..k+3, half 1 holds k+4..k+7; MFMA i
// of the block consumes element i of both halves, half 0 first).
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTile = 128;  // block tile: 128 queries x 128 candidates, 4 waves as 2x2 of 64x64

__device__ __forceinline__ float chain_dot(const float* __restrict__ q, const float* __restrict__ e, int K, int Kpad) {
  float acc = 0.f;
  // rows of K % 4 == 0 floats on 16-byte boundaries (every table the sweeps see): 128-bit loads, the next block of 8 in flight
  // behind the 8 dependent FMAs of the current one.  Same operations in the same order as the scalar loop below.
  if ((K & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(e)) & 15) == 0) {
    const float4* q4 = reinterpret_cast<const float4*>(q);
    const float4* e4 = reinterpret_cast<const float4*>(e);
    const int K4 = K >> 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 qa = q4[0], qb = q4[1], ea = e4[0], eb = 1 < K4 ? e4[1] : z;
    for (int b = 0; b < Kpad / 8; ++b) {
      float4 nqa = z, nqb = z, nea = z, neb = z;
      if (b + 1 < Kpad / 8) {
        nqa = q4[2 * b + 2]; nqb = q4[2 * b + 3];
        nea = 2 * b + 2 < K4 ? e4[2 * b + 2] : z;
        neb = 2 * b + 3 < K4 ? e4[2 * b + 3] : z;
      }
      acc = __fmaf_rn(qa.x, ea.x, acc); acc = __fmaf_rn(qb.x, eb.x, acc);
      acc = __fmaf_rn(qa.y, ea.y, acc); acc = __fmaf_rn(qb.y, eb.y, acc);
      acc = __fmaf_rn(qa.z, ea.z, acc); acc = __fmaf_rn(qb.z, eb.z, acc);
      acc = __fmaf_rn(qa.w, ea.w, acc); acc = __fmaf_rn(qb.w, eb.w, acc);
      qa = nqa; qb = nqb; ea = nea; eb = neb;
    }
    return acc;
  }
  for (int k0 = 0; k0 < Kpad; k0 += 8) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ka = k0 + i, kb = k0 + 4 + i;
      acc = __fmaf_rn(q[ka], ka < K ? e[ka] : 0.f, acc);
      acc = __fmaf_rn(q[kb], kb < K ? e[kb] : 0.f, acc);
    }
  }
  return acc;
}

// side 0 = tail queries (s,p,?) ; side 1 = head queries (?,p,o)
// RotatE (models.py:439-459) is linear in s and in o, so a 1-vs-all query is a dot product too: the order of the candidates by
// score is their order by <q, e_c>, with  tails: q = +-1 (score = const + sum of the candidate's row; the reference's score ranks all
// tails of every query by their row sums),  heads: q = -+[cos p + sin p | cos p - sin p]; sign by gamma (gamma - S / S).
template <int MODEL>
__global__ __launch_bounds__(256) void rank_build_q_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                            const int32_t* __restrict__ triples, int64_t nq, int side, int N,
                                                            int R, int d_e, int Kpad, float* __restrict__ Qv,
                                                            uint32_t* __restrict__ err, float gamma = 0.f) {
  constexpr int G = 64;
  const int lane = threadIdx.x & (G - 1);
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (q >= nq) return;
  const int s = triples[3 * q + 0], p = triples[3 * q + 1], o = triples[3 * q + 2];
  const int ent = side == 0 ? s : o;
  const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
  const float* er = E + (int64_t)(ok ? ent : 0) * d_e;
  const float* rp = Rel + (int64_t)(ok ? p : 0) * (MODEL == KGE_ROTATE ? d_e >> 1 : d_e);
  float* out = Qv + q * Kpad;
  if (!ok && lane == 0) atomicOr(err, 1u);
  if (MODEL == KGE_ROTATE) {
    const int h = d_e >> 1;
    const float sg = gamma > 0.f ? 1.f : -1.f;
    for (int k = lane; k < Kpad; k += G) {
      float v = 0.f;
      if (ok && k < d_e) {
        if (side == 0) v = sg;
        else {
          float sn, c;
          sincosf(rp[k < h ? k : k - h], &sn, &c);
          v = -sg * (k < h ? c + sn : c - sn);
        }
      }
      out[k] = v;
    }
  } else if (MODEL == KGE_DISTMULT) {
    for (int k = lane; k < Kpad; k += G) out[k] = (ok && k < d_e) ? er[k] * rp[k] : 0.f;
  } else if (MODEL == KGE_TRANSE) {
    // ||s+p-o||: tail candidates o against q = s+p ; head candidates s against q = o-p  (|q-e| = |e-q|)
    for (int k = lane; k < Kpad; k += G) out[k] = (ok && k < d_e) ? (side == 0 ? er[k] + rp[k] : er[k] - rp[k]) : 0.f;
  } else {  // ComplEx, rows are [real | imag]
    const int h = d_e >> 1;
    for (int k = lane; k < Kpad; k += G) {
      float v = 0.f;
      if (ok && k < d_e) {
        const int kk = k < h ? k : k - h;
        const float xr = er[kk], xi = er[h + kk], pr = rp[kk], pi = rp[h + kk];
        if (side == 0) v = k < h ? xr * pr - xi * pi : xi * pr + xr * pi;      // coefficients of o_real | o_imag
        else v = k < h ? pr * xr + pi * xi : pr * xi - pi * xr;                // coefficients of s_real | s_imag
      }
      out[k] = v;
    }
  }
}

__global__ __launch_bounds__(256) void rank_target_kernel(const float* __restrict__ Qv, const float* __restrict__ E,
                                                           const int32_t* __restrict__ triples, int64_t nq, int side, int N,
                                                           int d_e, int Kpad, float* __restrict__ tgt,
                                                           int32_t* __restrict__ counts, int zero_counts) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int t = triples[3 * q + (side == 0 ? 2 : 0)];
  const bool ok = (unsigned)t < (unsigned)N;
  tgt[q] = ok ? chain_dot(Qv + q * Kpad, E + (int64_t)t * d_e, d_e, Kpad) : __builtin_nanf("");
  if (zero_counts) counts[q] = 0;
}

// Target scores for rows of K % 4 == 0 floats on 16-byte boundaries: one wave per 64 queries.  A thread per query alone
// (rank_target_kernel) is latency bound -- 1.25 waves per CU at 20 466 queries, every load instruction touching 64 different
// rows (63-75 us).  Here the wave copies the 64 query rows and the 64 target rows chunk by chunk into LDS with coalesced 128-bit
// loads (ten consecutive lanes read 160 contiguous bytes of one row), the next chunk in flight in registers behind the current
// one, and every lane then runs ITS query's chain from LDS -- the same fmaf sequence in the same k order as chain_dot().
template <int BK, bool L2 = false>
__global__ __launch_bounds__(64) void rank_target4_kernel(const float* __restrict__ Qv, const float* __restrict__ E,
                                                           const int32_t* __restrict__ triples, int nq, int side, int N, int K,
                                                           int Kpad, float* __restrict__ tgt, int32_t* __restrict__ counts,
                                                           int zero_counts, const float* __restrict__ en = nullptr) {
  constexpr int LS = BK + 4, C4 = BK / 4, PF = C4;  // 64 rows x C4 float4 per operand per chunk = C4 per lane
  __shared__ __attribute__((aligned(16))) float Aq[64 * LS];
  __shared__ __attribute__((aligned(16))) float Ae[64 * LS];
  __shared__ int trow[64];
  const int lane = threadIdx.x;
  const int q0 = blockIdx.x * 64;
  const int q = q0 + lane;
  int t = -1;
  if (q < nq) t = triples[3 * (int64_t)q + (side == 0 ? 2 : 0)];
  const bool ok = (unsigned)t < (unsigned)N;
  trow[lane] = ok ? t : 0;
  __syncthreads();
  const int last_q = nq - 1, nchunk = Kpad / BK;
  float4 pa[PF], pb[PF];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = lane + u * 64, r = idx / C4, c = (idx - r * C4) * 4, k = kc * BK + c;
      pa[u] = *reinterpret_cast<const float4*>(Qv + (int64_t)min(q0 + r, last_q) * Kpad + k);
      const bool in = k < K;  // K % 4 == 0: inside the row or entirely in the padding
      const float4 v = *reinterpret_cast<const float4*>(E + (int64_t)trow[r] * K + (in ? k : 0));
      pb[u] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(0);
  float acc = 0.f;
  for (int kc = 0; kc < nchunk; ++kc) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = lane + u * 64, r = idx / C4, c = (idx - r * C4) * 4;
      *reinterpret_cast<float4*>(Aq + r * LS + c) = pa[u];
      *reinterpret_cast<float4*>(Ae + r * LS + c) = pb[u];
    }
    __syncthreads();
    if (kc + 1 < nchunk) fetch(kc + 1);
#pragma unroll
    for (int b = 0; b < BK / 8; ++b) {
      const float4 qa = *reinterpret_cast<const float4*>(Aq + lane * LS + 8 * b), qb = *reinterpret_cast<const float4*>(Aq + lane * LS + 8 * b + 4);
      const float4 ea = *reinterpret_cast<const float4*>(Ae + lane * LS + 8 * b), eb = *reinterpret_cast<const float4*>(Ae + lane * LS + 8 * b + 4);
      acc = __fmaf_rn(qa.x, ea.x, acc); acc = __fmaf_rn(qb.x, eb.x, acc);
      acc = __fmaf_rn(qa.y, ea.y, acc); acc = __fmaf_rn(qb.y, eb.y, acc);
      acc = __fmaf_rn(qa.z, ea.z, acc); acc = __fmaf_rn(qb.z, eb.z, acc);
      acc = __fmaf_rn(qa.w, ea.w, acc); acc = __fmaf_rn(qb.w, eb.w, acc);
    }
  }
  if (q < nq) {
    // L2: the TransE-L2 comparison value D = |e|^2 - 2 <q, e> (see rank_sumsq_kernel)
    tgt[q] = ok ? (L2 ? __fmaf_rn(-2.f, acc, en[t]) : acc) : __builtin_nanf("");
    if (zero_counts) counts[q] = 0;
  }
}

// TransE with the L2 norm on the matrix cores: ||q - e||^2 = |q|^2 + |e|^2 - 2 <q, e>, and for one query the order of the
// candidates by distance is the order by D(q, e) = |e|^2 - 2 <q, e> (|q|^2 is common to all of them, the square root is monotone).
// D is formed with ONE fused multiply-add from the dot product -- the MFMA chain of the sweep, or chain_dot() in the target and
// filter kernels, bit-identical as for DistMult -- and the row's squared norm en[e], which this kernel computes once per call for
// the whole table; every comparison in the sweep, the target and the filter kernels is therefore made on bit-identical numbers.
// gamma > 0: score = gamma - dist, so "score_c > score_t" is D_c < D_t; gamma <= 0: score = +dist, D_c > D_t.
// One 16-lane group per row; lane l takes the float4 chunks l, l+16, ...; fixed order, so en[] is deterministic.
__global__ __launch_bounds__(256) void rank_sumsq_kernel(const float* __restrict__ E, int N, int K, float* __restrict__ en) {
  const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  if (g >= N) return;
  const float4* e4 = reinterpret_cast<const float4*>(E + g * K);
  float acc = 0.f;
  for (int c = l; c < (K >> 2); c += 16) {
    const float4 v = e4[c];
    acc = __fmaf_rn(v.x, v.x, acc); acc = __fmaf_rn(v.y, v.y, acc); acc = __fmaf_rn(v.z, v.z, acc); acc = __fmaf_rn(v.w, v.w, acc);
  }
  acc = group_sum<16>(acc);
  if (l == 0) en[g] = acc;
}

// Filter kernel of that path: un-counts a filtered candidate if the sweep counted it (same D, same comparison).
template <bool LESS>
__global__ __launch_bounds__(256) void rank_filter_l2_kernel(const float* __restrict__ Qv, const float* __restrict__ tgt,
                                                              const float* __restrict__ Ec, const float* __restrict__ en_c, int cand_offset,
                                                              int cand_rows, int K, int Kpad, const int32_t* __restrict__ pair_q,
                                                              const int32_t* __restrict__ filt_ids, int64_t npairs,
                                                              int32_t* __restrict__ counts, int q_lo = 0, int q_n = 0x7fffffff) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npairs) return;
  const int q = pair_q[e] - q_lo;
  if ((unsigned)q >= (unsigned)q_n) return;  // (a pair of the other side's queries: see rank_counts, side 2)
  const int c = filt_ids[e] - cand_offset;
  if ((unsigned)c >= (unsigned)cand_rows) return;  // another shard owns this candidate
  const float D = __fmaf_rn(-2.f, chain_dot(Qv + (int64_t)q * Kpad, Ec + (int64_t)c * K, K, Kpad), en_c[c]);
  if (LESS ? D < tgt[q] : D > tgt[q]) atomicSub(&counts[q], 1);
}

template <int BK, bool VEC4>
__global__ __launch_bounds__(256) void rank_sweep_kernel(const float* __restrict__ Qv, const float* __restrict__ tgt,
                                                          int64_t nq, const float* __restrict__ Ec, int K, int Kpad,
                                                          int cand_rows, int tiles_per_block, int32_t* __restrict__ counts) {
  constexpr int LS = BK + 4;         // LDS row stride in floats: (BK+4)/4 odd -> conflict-free ds_read_b128
  constexpr int C4 = BK / 4;         // float4 per tile row
  __shared__ __attribute__((aligned(16))) float As[kTile * LS];
  __shared__ __attribute__((aligned(16))) float Bs[kTile * LS];
  __shared__ int cnt_s[kTile];
  __shared__ float tg_s[kTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;
  const int64_t q0 = (int64_t)blockIdx.x * kTile;
  const int nct = (cand_rows + kTile - 1) / kTile;
  const int ct_begin = blockIdx.y * tiles_per_block;
  const int ct_end = min(nct, ct_begin + tiles_per_block);

  // Register budget (2 waves per SIMD need <= 256): the per-row targets live in LDS (broadcast reads in the epilogue)
  // and the 32 per-lane counters are packed four to a register (a lane adds at most 2 per candidate tile, and
  // tiles_per_block <= 60, so a byte never overflows).
  uint32_t cntp[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int w = 0; w < 4; ++w) cntp[rt][w] = 0u;
  if (tid < kTile) {
    cnt_s[tid] = 0;
    tg_s[tid] = q0 + tid < nq ? tgt[q0 + tid] : __builtin_inff();  // nothing is > +inf; NaN targets count nothing either
  }

  // Software pipeline over the flattened (candidate tile, k chunk) sequence: the global loads of step it+1 are issued
  // into registers before the MFMAs of step it and written to LDS after them, so L2 latency hides under matrix work.
  constexpr int PF = (kTile * C4 + 255) / 256;  // float4 per thread per operand per chunk
  const int nchunk = Kpad / BK;
  const int nsteps = (ct_end - ct_begin) * nchunk;
  float4 pa[PF], pb[PF];
  auto prefetch = [&](int it) {
    const int c0 = (ct_begin + it / nchunk) * kTile, k0 = (it % nchunk) * BK;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + u * 256;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      pb[u] = pa[u];
      if (idx < kTile * C4) {
        const int r = idx / C4, c4 = idx - r * C4;
        const int k = k0 + c4 * 4;
        if (q0 + r < nq) pa[u] = *reinterpret_cast<const float4*>(Qv + (q0 + r) * Kpad + k);  // Qv is zero padded to Kpad
        const int cr = c0 + r;
        if (cr < cand_rows) {
          const float* src = Ec + (int64_t)cr * K + k;
          if (VEC4) {
            if (k < K) pb[u] = *reinterpret_cast<const float4*>(src);  // K % 4 == 0
          } else {
            if (k + 0 < K) pb[u].x = src[0];
            if (k + 1 < K) pb[u].y = src[1];
            if (k + 2 < K) pb[u].z = src[2];
            if (k + 3 < K) pb[u].w = src[3];
          }
        }
      }
    }
  };
  f32x16 acc[2][2];
  if (nsteps > 0) prefetch(0);
  for (int it = 0; it < nsteps; ++it) {
    const int kc = it % nchunk;
    if (kc == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    __syncthreads();  // everyone is done reading the previous chunk
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + u * 256;
      if (idx < kTile * C4) {
        const int r = idx / C4, c4 = idx - r * C4;
        *reinterpret_cast<float4*>(As + r * LS + c4 * 4) = pa[u];
        *reinterpret_cast<float4*>(Bs + r * LS + c4 * 4) = pb[u];
      }
    }
    __syncthreads();
    if (it + 1 < nsteps) prefetch(it + 1);
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float4 a[2], b[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        a[x] = *reinterpret_cast<const float4*>(As + (wq * 64 + x * 32 + l31) * LS + 8 * j + 4 * half);
        b[x] = *reinterpret_cast<const float4*>(Bs + (wc * 64 + x * 32 + l31) * LS + 8 * j + 4 * half);
      }
#define KGE_STEP(comp)                                                                                         \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].comp, b[0].comp, acc[0][0], 0, 0, 0);                  \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].comp, b[1].comp, acc[0][1], 0, 0, 0);                  \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1].comp, b[0].comp, acc[1][0], 0, 0, 0);                  \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1].comp, b[1].comp, acc[1][1], 0, 0, 0);
      KGE_STEP(x) KGE_STEP(y) KGE_STEP(z) KGE_STEP(w)
#undef KGE_STEP
    }
    if (kc == nchunk - 1) {
      // fused count-greater epilogue: D[i][j], j = lane&31 (candidate), i = (r&3) + 8*(r>>2) + 4*half (query)
      const int c0 = (ct_begin + it / nchunk) * kTile;
      const bool last = c0 + kTile > cand_rows;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float tgv = tg_s[wq * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const bool valid = !last || (c0 + wc * 64 + cb * 32 + l31 < cand_rows);
            cntp[rt][r >> 2] += (valid && acc[rt][cb][r] > tgv) ? (1u << (8 * (r & 3))) : 0u;
          }
        }
    }
  }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wq * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int c = (int)((cntp[rt][r >> 2] >> (8 * (r & 3))) & 0xFFu);
      if (c) atomicAdd(&cnt_s[row], c);
    }
  __syncthreads();
  if (tid < kTile && q0 + tid < nq && cnt_s[tid]) atomicAdd(&counts[q0 + tid], cnt_s[tid]);
}

// The matrix-core sweep for rows of K % 4 == 0 floats on 16-byte boundaries (every table the models hand over): 256 x 128 block tiles,
// ONE 512-thread workgroup per CU (8 waves as 4 x 2 of 64 x 64, two per SIMD), PERSISTENT over a contiguous range of (query tile,
// candidate tile) units, the k chunks double-buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4: global memory -> LDS without
// passing through registers; no LDS store instruction at all: a wave issues 9 DMA loads per chunk, one barrier per chunk).  What the
// pipe can deliver to this MFMA stream was measured first (tools/mfma_probe.cpp: 96-98 % of its peak, fragments from LDS and barriers
// included); what a sweep loses is (a) instruction bursts in front of MFMA groups -- the two waves of a SIMD run in step, so a burst in
// one is a burst in both and the pipe idles for its length -- and (b) per-unit work that nothing overlaps.  Hence:
//   * the addresses of the DMA loads are a wave-uniform base pointer + a 32-bit offset per lane (three vector instructions per load
//     and chunk), and every non-MFMA instruction of a group of MFMAs is interleaved with them (sched_group_barrier);
//   * round 4: the LAST chunk of a unit runs TILE-MAJOR -- all of its k steps on accumulator tile (0,0), then (0,1), (1,0), (1,1) --
//     so that the count-greater epilogue of a finished 32 x 32 tile (16 compares + 16 add-with-carry per lane) rides between the MFMAs
//     of the next tile instead of standing behind the unit's last MFMA in both waves of every SIMD at once (four vector compares per
//     MFMA slot, against a pipe that idled for the whole epilogue: ~2 000 of a K = 200 unit's 51 200 pipe cycles); only the epilogue of
//     tile (1,1) is left behind the last MFMA.  A tile's own k order does not change -- an accumulator sees its chunk's k steps in the
//     same ascending order whether the other tiles' steps sit between them or not -- so every accumulation chain is still the one
//     documented at the top of the file and the target / filter kernels (chain_dot) see bit-identical scores.
// LDS-DMA writes 64 lanes x 16 bytes = one contiguous KiB per instruction ("segment"); the images keep their padded rows (BK + 4
// floats: conflict-free ds_read_b128 fragments), so lane l of segment g fills float4 number 64 g + l of the image -- row (64 g + l) /
// (C4 + 1), column (64 g + l) % (C4 + 1) -- from wherever that float4 lives in global memory; the pad column repeats column 0 (never
// read).  Segments are dealt to the waves round-robin; where the count does not divide by 8 the last slot of some waves wraps around and
// repeats a segment (the same bytes to the same place) rather than being predicated: a branch would cut the chunk's instruction stream
// into separately scheduled pieces.  TAIL (K < Kpad): candidate float4s past K come from the zero padding of the first query row.
constexpr int kQT5 = 256;
#ifdef KGE_SWEEP5_STAMP  // diagnostic build (tools/sweep_stamps.py): wall-clock stamp (100 MHz) of every unit end, per workgroup
constexpr int kStampUnits = 512;
__device__ unsigned long long g_sweep5_stamps[256 * kStampUnits];
#endif
template <int BK, bool TAIL, int EPI = 0>
__global__ __launch_bounds__(512) void rank_sweep5_kernel(const float* __restrict__ Qv, const float* __restrict__ tgt,
                                                           int nq, const float* __restrict__ Ec, int K, int Kpad,
                                                           int cand_rows, int32_t* __restrict__ counts,
                                                           const float* __restrict__ en_c = nullptr, int parts = 0) {
  constexpr int LS = BK + 4;
  constexpr int C4 = BK / 4, S4 = C4 + 1;         // float4s per row: data, data + pad
  constexpr int G = BK / 8;                       // groups of 16 MFMAs per chunk
  constexpr int SEG_A = kQT5 * S4 / 64, SEG_B = kTile * S4 / 64;   // KiB segments of the two images: 44 / 22 (BK 40), 36 / 18 (BK 32)
  constexpr int SA = (SEG_A + 7) / 8, SB = (SEG_B + 7) / 8;        // DMA slots of a wave: 6 + 3 (BK 40), 5 + 3 (BK 32)
  constexpr int NS = SA + SB;
  static_assert(kQT5 * S4 % 64 == 0 && kTile * S4 % 64 == 0, "images are whole segments");
  static_assert(NS <= 9 && G >= 4, "slots 0..8, issued in groups 0..2");
  __shared__ __attribute__((aligned(1024))) float As[2][kQT5 * LS];
  __shared__ __attribute__((aligned(1024))) float Bs[2][kTile * LS];
  __shared__ int cnt_s[kQT5];
  __shared__ __attribute__((aligned(16))) float tg_s[kQT5];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;
  const int nqt = (nq + kQT5 - 1) / kQT5, nct = (cand_rows + kTile - 1) / kTile;
  const int64_t total = (int64_t)nqt * nct;
  const int nb = gridDim.x;
  // workgroups are dealt round-robin to the 8 XCDs: logical block lb makes the workgroups of an XCD neighbours in the unit order
  const int lb = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  int64_t u_begin, u_end;
  if (parts > 0) {  // phase-aligned cut: every workgroup sweeps 1/parts of ONE query tile's candidate tiles, all workgroups walking in step
    const int q_ = lb / parts, p_ = lb - q_ * parts;
    u_begin = (int64_t)q_ * nct + (int64_t)nct * p_ / parts;
    u_end = (int64_t)q_ * nct + (int64_t)nct * (p_ + 1) / parts;
  } else {
    u_begin = total * lb / nb;
    u_end = total * (lb + 1) / nb;
  }
  if (u_begin >= u_end) return;
  // (the tile coordinates index scalar base pointers: say so to the compiler, which computes the 64-bit division in vector registers)
  int qt = __builtin_amdgcn_readfirstlane((int)(u_begin / nct));
  int ct = __builtin_amdgcn_readfirstlane((int)(u_begin - (int64_t)qt * nct));
  int q0 = qt * kQT5;

  int cnt[2][16];  // one counter per accumulator row (both candidate halves of the wave tile add into it)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) cnt[rt][r] = 0;
  if (tid < kQT5) {
    cnt_s[tid] = 0;
    tg_s[tid] = q0 + tid < nq ? tgt[q0 + tid] : (EPI == 1 ? -__builtin_inff() : __builtin_inff());
  }

  // DMA slots as named scalars (slot U < SA fills a segment of the query image, the others of the candidate image; slots past NS are
  // compiled out): sg = the segment; rb = byte offset of this lane's float4 from the tile's base pointer, cb = the part of it inside
  // the row.  Rows past the tile's last valid row are clamped to that row (queries past nq never store their counters, candidates past
  // cand_rows are masked in the epilogue): of = min(rb, last valid row's offset + cb), two vector instructions per slot and chunk.
  const int last_row = cand_rows - 1, last_q = nq - 1;
  const int nchunk = Kpad / BK;  // >= 2 (the launcher sends rows of a single chunk, K <= BK, to the generic kernel): chunk 0 opens the accumulators
#define KGE5_SLOT(U)                                                                                     \
  const int sg##U = U < SA ? (wave + 8 * U) % SEG_A : (wave + 8 * (U - SA)) % SEG_B;                     \
  const int f##U = sg##U * 64 + lane;                                                                    \
  const int r##U = f##U / S4, c##U = f##U - r##U * S4;                                                   \
  const unsigned cb##U = (unsigned)(c##U == C4 ? 0 : c##U) * 16u;                                        \
  const unsigned rb##U = (unsigned)r##U * (unsigned)(U < SA ? Kpad : K) * 4u + cb##U;                    \
  unsigned of##U = 0;
  KGE5_SLOT(0) KGE5_SLOT(1) KGE5_SLOT(2) KGE5_SLOT(3) KGE5_SLOT(4) KGE5_SLOT(5) KGE5_SLOT(6) KGE5_SLOT(7) KGE5_SLOT(8)
#undef KGE5_SLOT
  // limA / limB: byte offset of the last valid row of the query / candidate tile (uniform)
#define KGE5_OFF(U)                                                                                      \
  if (U < NS) of##U = min(rb##U, (U < SA ? limA : limB) + cb##U);
  // gA / gB: base pointers of the two tiles at the chunk's first k (TAIL: gB at k = 0, kB_ = the chunk's first k); BUF: the image.
  // The DMA loads are inline assembly: issued through the compiler's builtin, every later ds_read of the OTHER image would wait for
  // them (the compiler cannot tell the two images apart: s_waitcnt vmcnt(0) in front of each group's fragment reads).  What the
  // compiler therefore does not do is done here: M0 (the LDS destination of the wave's 64 x 16 bytes) is written in the statement
  // that reads it, and the loads are waited for by an explicit s_waitcnt vmcnt(0) in front of the chunk's barrier.  A load whose
  // address is a scalar base + a 32-bit byte offset per lane takes the saddr form.
  const unsigned lds_a = (unsigned)(uintptr_t)&As[0][0], lds_b = (unsigned)(uintptr_t)&Bs[0][0];
  unsigned keep_m0;
#define KGE5_DMA_S(BASE, BOFF, LDS)                                                                                           \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"        \
               : "=&s"(keep_m0) : "v"(BOFF), "s"(LDS), "s"(BASE) : "memory")
#define KGE5_DMA_V(PTR, LDS)                                                                                                  \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"       \
               : "=&s"(keep_m0) : "v"(PTR), "s"(LDS) : "memory")
#define KGE5_LD(U, BUF)                                                                                  \
  if (U < NS) {                                                                                          \
    if (U < SA) {                                                                                        \
      KGE5_DMA_S(gA, of##U, lds_a + (unsigned)(((BUF) * (kQT5 * LS) + sg##U * 256) * 4));           \
    } else if (TAIL) { /* K % 4 == 0: a float4 is inside the row or entirely in the padding */           \
      const float* s_ = kB_ + (int)(cb##U >> 2) < K ? gB + ((of##U >> 2) + (unsigned)kB_) : Qv + K;                         \
      KGE5_DMA_V(s_, lds_b + (unsigned)(((BUF) * (kTile * LS) + sg##U * 256) * 4));                      \
    } else {                                                                                             \
      KGE5_DMA_S(gB, of##U, lds_b + (unsigned)(((BUF) * (kTile * LS) + sg##U * 256) * 4));          \
    }                                                                                                    \
  }
  // next chunk into the other image: three DMA loads (offset, then load) in each of the first three groups of a chunk
#define KGE5_STAGE(J, BUF)                                                                                                   \
  if ((J) == 0) { KGE5_OFF(0) KGE5_OFF(1) KGE5_OFF(2) KGE5_LD(0, BUF) KGE5_LD(1, BUF) KGE5_LD(2, BUF) }                        \
  if ((J) == 1) { KGE5_OFF(3) KGE5_OFF(4) KGE5_OFF(5) KGE5_LD(3, BUF) KGE5_LD(4, BUF) KGE5_LD(5, BUF) }                        \
  if ((J) == 2) { KGE5_OFF(6) KGE5_OFF(7) KGE5_OFF(8) KGE5_LD(6, BUF) KGE5_LD(7, BUF) KGE5_LD(8, BUF) }
  // the instruction stream of one group: N x { one MFMA, at most one memory instruction (LDS read / DMA load), at most ALU vector /
  // scalar instructions } -- fragment reads, DMA loads, address arithmetic and (last chunk) the epilogue ride between the MFMAs
#define KGE5_PIPE(N, MEM, ALU)                                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < (N); ++i_) {                                                   \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100 | 0x010, (MEM), 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x002 | 0x004, (ALU), 0);                                       \
  }
  const int a_off = (wq * 64 + l31) * LS + 4 * half;
  const int b_off = (wc * 64 + l31) * LS + 4 * half;

  // prologue: the first chunk of the first unit into image 0
  {
    const int c0_ = ct * kTile;
    const float* gA = Qv + (int64_t)q0 * Kpad;
    const float* gB = Ec + (int64_t)c0_ * K;
    const int kB_ = 0;
    (void)kB_;
    const unsigned limA = (unsigned)min(last_q - q0, kQT5 - 1) * (unsigned)Kpad * 4u, limB = (unsigned)min(last_row - c0_, kTile - 1) * (unsigned)K * 4u;
    KGE5_OFF(0) KGE5_OFF(1) KGE5_OFF(2) KGE5_OFF(3) KGE5_OFF(4) KGE5_OFF(5) KGE5_OFF(6) KGE5_OFF(7) KGE5_OFF(8)
    KGE5_LD(0, 0) KGE5_LD(1, 0) KGE5_LD(2, 0) KGE5_LD(3, 0) KGE5_LD(4, 0) KGE5_LD(5, 0) KGE5_LD(6, 0) KGE5_LD(7, 0) KGE5_LD(8, 0)
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int p = 0;  // image the running chunk is read from
  for (int64_t u = u_begin; u < u_end; ++u) {
    // coordinates of the next unit of this workgroup's range (after the last one: this unit again, its values are dropped)
    const bool has_next = u + 1 < u_end;
    int nqt_ = qt, nct_ = ct;
    if (has_next) {
      nct_ = ct + 1;
      if (nct_ == nct) { nct_ = 0; nqt_ = qt + 1; }
    }
    f32x16 acc[2][2];  // (no zero fill: the first MFMA of every accumulator takes a zero C operand -- 64 vector moves per wave and unit less)
    // ---- chunks 0 .. nchunk - 2, group-major: a group = one float4 of k per operand row = 4 k-steps x 4 accumulator tiles
    for (int kc = 0; kc + 1 < nchunk; ++kc) {
      // every wave's DMA loads of image p have landed (issued in the first three groups of the previous chunk); everyone is done
      // reading image p ^ 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const float* a_rd = &As[p][a_off];
      const float* b_rd = &Bs[p][b_off];
      const int nk0 = (kc + 1) * BK;  // the chunk staged next: the next chunk of this unit
      const float* gA = Qv + (int64_t)q0 * Kpad + nk0;
      const float* gB = Ec + (int64_t)(ct * kTile) * K + (TAIL ? 0 : nk0);
      const int kB_ = nk0;
      (void)kB_;
      const unsigned limA = (unsigned)min(last_q - q0, kQT5 - 1) * (unsigned)Kpad * 4u, limB = (unsigned)min(last_row - ct * kTile, kTile - 1) * (unsigned)K * 4u;
      float4 fa[2][2], fb[2][2];  // [buffer][row tile]: the fragments of group j+1 are read while group j multiplies
#define KGE_FRAG(BUF, J)                                                                          \
  _Pragma("unroll") for (int x = 0; x < 2; ++x) {                                                 \
    fa[BUF][x] = *reinterpret_cast<const float4*>(a_rd + x * 32 * LS + 8 * (J));                  \
    fb[BUF][x] = *reinterpret_cast<const float4*>(b_rd + x * 32 * LS + 8 * (J));                  \
  }
#define KGE_STEP(BUF, comp)                                                                                           \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][0].comp, fb[BUF][0].comp, acc[0][0], 0, 0, 0);             \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][0].comp, fb[BUF][1].comp, acc[0][1], 0, 0, 0);             \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][1].comp, fb[BUF][0].comp, acc[1][0], 0, 0, 0);             \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][1].comp, fb[BUF][1].comp, acc[1][1], 0, 0, 0);
#define KGE_STEP0(BUF, comp)                                                                                      \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][0].comp, fb[BUF][0].comp, zero16, 0, 0, 0);               \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][0].comp, fb[BUF][1].comp, zero16, 0, 0, 0);               \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][1].comp, fb[BUF][0].comp, zero16, 0, 0, 0);               \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][1].comp, fb[BUF][1].comp, zero16, 0, 0, 0);
      KGE_FRAG(0, 0)
#pragma unroll
      for (int j = 0; j < G; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < G) {
          if ((j & 1) == 0) { KGE_FRAG(1, j + 1) } else { KGE_FRAG(0, j + 1) }
        }
        KGE5_STAGE(j, p ^ 1)
        if (j == 0) {
          if (kc == 0) { KGE_STEP0(0, x) } else { KGE_STEP(0, x) }
          KGE_STEP(0, y) KGE_STEP(0, z) KGE_STEP(0, w)
        } else if ((j & 1) == 0) { KGE_STEP(0, x) KGE_STEP(0, y) KGE_STEP(0, z) KGE_STEP(0, w) }
        else { KGE_STEP(1, x) KGE_STEP(1, y) KGE_STEP(1, z) KGE_STEP(1, w) }
        KGE5_PIPE(16, 1, 2)
      }
      __builtin_amdgcn_sched_barrier(0);
#undef KGE_STEP
#undef KGE_STEP0
#undef KGE_FRAG
      p ^= 1;
    }
    // ---- the unit's last chunk, tile-major, with the count-greater epilogue of tile t - 1 between the MFMAs of tile t.
    // D[i][j] of a 32 x 32 tile: j = lane & 31 (candidate), i = (r & 3) + 8 * (r >> 2) + 4 * half (query), r = accumulator register.
    const int c0 = ct * kTile;
    // (this lane's two candidate columns exist: false only in the table's last, partial tile -- its rows past the end are clamped copies
    // of the last row and must not count)
    const bool v0 = c0 + wc * 64 + l31 < cand_rows, v1 = c0 + wc * 64 + 32 + l31 < cand_rows;
    float en0 = 0.f, en1 = 0.f;  // squared norms of this lane's two candidate columns (EPI != 0)
    if (EPI != 0) {
      en0 = en_c[min(c0 + wc * 64 + l31, last_row)];
      en1 = en_c[min(c0 + wc * 64 + 32 + l31, last_row)];
    }
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const float* a_rd = &As[p][a_off];
      const float* b_rd = &Bs[p][b_off];
      // the chunk staged next: the first chunk of the next unit
      const int nq0 = nqt_ * kQT5, nc0 = nct_ * kTile;
      const float* gA = Qv + (int64_t)nq0 * Kpad;
      const float* gB = Ec + (int64_t)nc0 * K;
      const int kB_ = 0;
      (void)kB_;
      const unsigned limA = (unsigned)min(last_q - nq0, kQT5 - 1) * (unsigned)Kpad * 4u, limB = (unsigned)min(last_row - nc0, kTile - 1) * (unsigned)K * 4u;
      float4 ta[2], tb[2];   // [buffer]: the fragment pair of the tile's next group is read while this group multiplies
      float4 tg4[4];         // targets of the row tile's 16 accumulator rows of this lane: tg4[q].{x,y,z,w} = rows 8 q + 4 half + {0..3}
#define KGE_TFRAG(BUF, T, J)                                                                                  \
  ta[BUF] = *reinterpret_cast<const float4*>(a_rd + ((T) >> 1) * 32 * LS + 8 * (J));                          \
  tb[BUF] = *reinterpret_cast<const float4*>(b_rd + ((T) & 1) * 32 * LS + 8 * (J));
#define KGE_TSTEP(T, BUF, comp) \
  acc[(T) >> 1][(T) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[BUF].comp, tb[BUF].comp, acc[(T) >> 1][(T) & 1], 0, 0, 0);
      // epilogue of accumulator row R of tile T (row tile T >> 1, candidate half T & 1): counter += (score > target), as a vector compare
      // into VCC and an add-with-carry -- two vector instructions per score.  (Written out: from `cnt += hit ? 1 : 0` the compiler makes
      // compare, wait state, select 0 / 1 into a fresh register, add.)  No column mask here: see the partial tile below.
#define KGE_TEPI1(T, R)                                                                                       \
  {                                                                                                           \
    const float4 t4_ = tg4[(R) >> 2];                                                                         \
    const float tgv = ((R) & 3) == 0 ? t4_.x : (((R) & 3) == 1 ? t4_.y : (((R) & 3) == 2 ? t4_.z : t4_.w));    \
    float sc_ = acc[(T) >> 1][(T) & 1][R];                                                                    \
    if (EPI != 0) sc_ = __fmaf_rn(-2.f, sc_, ((T) & 1) ? en1 : en0);                                          \
    if (EPI == 1) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(cnt[(T) >> 1][R]) : "v"(sc_), "v"(tgv) : "vcc"); \
    else asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(cnt[(T) >> 1][R]) : "v"(sc_), "v"(tgv) : "vcc");           \
  }
#define KGE_TEPI(T, R0, R1) _Pragma("unroll") for (int r = (R0); r < (R1); ++r) KGE_TEPI1(T, r)
#define KGE_TTARGETS(RT)                                                                                      \
  _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                               \
    tg4[q] = *reinterpret_cast<const float4*>(&tg_s[wq * 64 + (RT) * 32 + 8 * q + 4 * half]);
      KGE_TFRAG(0, 0, 0)
      KGE_TTARGETS(0)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int g = t * G + j;  // running group number: buffers alternate across the tile switch too
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < G) {
            if ((g & 1) == 0) { KGE_TFRAG(1, t, j + 1) } else { KGE_TFRAG(0, t, j + 1) }
          } else if (t + 1 < 4) {
            if ((g & 1) == 0) { KGE_TFRAG(1, t + 1, 0) } else { KGE_TFRAG(0, t + 1, 0) }
          }
          if (t == 0) { KGE5_STAGE(j, p ^ 1) }
          // the 16 accumulator rows of the finished tile t - 1 are dealt over this tile's G groups, one row behind each MFMA
          // (row e0 + i behind the group's MFMA i; the inline assembly is not an instruction class sched_group_barrier could place)
          const int e0 = j * 16 / G, e1 = (j + 1) * 16 / G;
#define KGE_TE(I) if (t > 0 && e0 + (I) < e1) KGE_TEPI1(t > 0 ? t - 1 : 0, e0 + (I) < 16 ? e0 + (I) : 15)
          if ((g & 1) == 0) { KGE_TSTEP(t, 0, x) KGE_TE(0) KGE_TSTEP(t, 0, y) KGE_TE(1) KGE_TSTEP(t, 0, z) KGE_TE(2) KGE_TSTEP(t, 0, w) KGE_TE(3) }
          else { KGE_TSTEP(t, 1, x) KGE_TE(0) KGE_TSTEP(t, 1, y) KGE_TE(1) KGE_TSTEP(t, 1, z) KGE_TE(2) KGE_TSTEP(t, 1, w) KGE_TE(3) }
#undef KGE_TE
        }
        if (t == 2) {  // the epilogues still to come (tile 2 during tile 3, tile 3 at the end) are the wave's second row tile: its targets
                       // (tile 1's epilogue, just issued between tile 2's MFMAs, was the last user of the first row tile's)
          __builtin_amdgcn_sched_barrier(0);
          KGE_TTARGETS(1)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      KGE_TEPI(3, 0, 16)
#undef KGE_TFRAG
#undef KGE_TSTEP
#undef KGE_TEPI
#undef KGE_TTARGETS
    }
    p ^= 1;
    if (c0 + kTile > cand_rows) {
      // (uniform; the table's last candidate tile, one unit in nct) the tile's rows past the end were filled with clamped copies of the
      // last row and the epilogue above counted them like any other column: the lanes that hold such columns take their hits back.  The
      // accumulators are still live (the next unit's first MFMAs overwrite them).
      auto hit = [&](float dot, float en, float tgv) {
        if (EPI == 0) return dot > tgv;
        const float D = __fmaf_rn(-2.f, dot, en);
        return EPI == 1 ? D < tgv : D > tgv;
      };
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float tgv = tg_s[wq * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
          cnt[rt][r] -= ((!v0 && hit(acc[rt][0][r], en0, tgv)) ? 1 : 0) + ((!v1 && hit(acc[rt][1][r], en1, tgv)) ? 1 : 0);
        }
    }
    if (!has_next || nqt_ != qt) {  // uniform: the range leaves this query tile -> flush its counters
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wq * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (cnt[rt][r]) atomicAdd(&cnt_s[row], cnt[rt][r]);
          cnt[rt][r] = 0;
        }
      __syncthreads();
      if (tid < kQT5) {
        if (q0 + tid < nq && cnt_s[tid]) atomicAdd(&counts[q0 + tid], cnt_s[tid]);
        if (has_next) {  // (read again only after the barrier at the top of the next chunk)
          const int nq0_ = nqt_ * kQT5;
          cnt_s[tid] = 0;
          tg_s[tid] = nq0_ + tid < nq ? tgt[nq0_ + tid] : (EPI == 1 ? -__builtin_inff() : __builtin_inff());
        }
      }
    }
    qt = nqt_;
    ct = nct_;
    q0 = qt * kQT5;
#ifdef KGE_SWEEP5_STAMP
    if (tid == 0 && blockIdx.x < 256 && u - u_begin < kStampUnits - 1) {
      if (u == u_begin) g_sweep5_stamps[blockIdx.x * kStampUnits] = (unsigned long long)(u_end - u_begin);
      g_sweep5_stamps[blockIdx.x * kStampUnits + 1 + (u - u_begin)] = __builtin_amdgcn_s_memrealtime();
    }
#endif
  }
#undef KGE5_LD
#undef KGE5_DMA_S
#undef KGE5_DMA_V
#undef KGE5_OFF
#undef KGE5_PIPE
#undef KGE5_STAGE
}

// One thread per CSR entry (query pair_q[e], filtered global id filt_ids[e]); un-counts it if it was counted.
__global__ __launch_bounds__(256) void rank_filter_kernel(const float* __restrict__ Qv, const float* __restrict__ tgt,
                                                           const float* __restrict__ Ec, int cand_offset, int cand_rows,
                                                           int K, int Kpad, const int32_t* __restrict__ pair_q,
                                                           const int32_t* __restrict__ filt_ids, int64_t npairs,
                                                           int32_t* __restrict__ counts, int q_lo = 0, int q_n = 0x7fffffff) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npairs) return;
  const int q = pair_q[e] - q_lo;
  if ((unsigned)q >= (unsigned)q_n) return;
  const int c = filt_ids[e] - cand_offset;
  if ((unsigned)c >= (unsigned)cand_rows) return;  // another shard owns this candidate
  const float sc = chain_dot(Qv + (int64_t)q * Kpad, Ec + (int64_t)c * K, K, Kpad);
  if (sc > tgt[q]) atomicSub(&counts[q], 1);
}

// ---------------------------------------------------------------------------------------------------------------
// TransE (models.py:186-190): score = gamma - ||q - e||_n (or +||.|| when gamma <= 0).  No dot product -> no MFMA: a
// register-tiled VALU sweep (each thread 4 queries x 4 candidates, operands through k-major LDS tiles), 2 VALU ops
// per (query, candidate, k).  Target and filtered-candidate scores come from dist_chain(), the same operation
// sequence in the same ascending-k order, so the strict comparisons are made on bit-identical numbers.
template <int NORM>
__device__ __forceinline__ float dist_step(float acc, float q, float e, float pw = 2.f) {
  const float x = q - e;
  return NORM == 1 ? acc + fabsf(x) : (NORM == 3 ? acc + phase_abs(x) : (NORM == 5 ? acc + powf(fabsf(x), pw) : __fmaf_rn(x, x, acc)));  // 5: general p-norm
}
// NORM 4 = the pRotatE phase distance (NORM 3) on rows that hold (sin(x/2), cos(x/2)) PAIRS: |sin((q - e)/2)| =
// |sin(q/2) cos(e/2) - cos(q/2) sin(e/2)|, one multiply, one fused multiply-add and one |.|-add per element instead of a sinf
// (about 30 VALU instructions with its range reduction).  The sweep, the target and the filter kernels all take this form (the
// tables are built once per call by rank_phase_table_kernel / rank_build_q_phase_kernel), so comparisons stay on bit-identical
// numbers; against the direct form the element differs by a few ulp.
__device__ __forceinline__ float phase_pair_step(float acc, float qs, float qc, float es, float ec) {
  return acc + fabsf(__fmaf_rn(-qc, es, __fmul_rn(qs, ec)));
}
template <int NORM>
__device__ __forceinline__ float dist_score(float acc, float gamma, float modulus) {
  const float d = NORM == 1 ? acc : (NORM == 5 ? powf(acc, 1.f / modulus) : (NORM >= 3 ? modulus * acc : sqrtf(acc)));  // 3, 4: pRotatE phase distance (models.py:479-489); 5: (sum |x|^p)^(1/p), p = modulus
  return gamma > 0.f ? gamma - d : d;
}
template <int NORM>
__device__ __forceinline__ float dist_chain(const float* __restrict__ q, const float* __restrict__ e, int K, float gamma, float modulus) {
  float acc = 0.f;
  if (NORM == 4) {
    for (int k = 0; k + 1 < K; k += 2) acc = phase_pair_step(acc, q[k], q[k + 1], e[k], e[k + 1]);
  } else {
    for (int k = 0; k < K; ++k) acc = dist_step<NORM>(acc, q[k], e[k], modulus);
  }
  return dist_score<NORM>(acc, gamma, modulus);
}

// Rows of (sin(x/2), cos(x/2)) pairs of a table (rows, K) -> (rows, 2K)
__global__ __launch_bounds__(256) void rank_phase_table_kernel(const float* __restrict__ E, int64_t n, float* __restrict__ P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sn, cs;
  sincosf(0.5f * E[i], &sn, &cs);
  reinterpret_cast<float2*>(P)[i] = make_float2(sn, cs);
}
// Query rows of the same form: q = s + p (tail side) or o - p (head side), zero padded to Kpad2 floats
__global__ __launch_bounds__(256) void rank_build_q_phase_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                                  const int32_t* __restrict__ triples, int64_t nq, int side, int N, int R,
                                                                  int K, int Kpad2, float* __restrict__ Qp, uint32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= nq) return;
  const int s = triples[3 * q + 0], p = triples[3 * q + 1], o = triples[3 * q + 2];
  const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
  if (!ok && lane == 0) atomicOr(err, 1u);
  const float* er = E + (int64_t)(ok ? (side == 0 ? s : o) : 0) * K;
  const float* rp = Rel + (int64_t)(ok ? p : 0) * K;
  float* out = Qp + q * Kpad2;
  for (int k = lane; k < Kpad2 / 2; k += 64) {
    float sn = 0.f, cs = 0.f;
    if (ok && k < K) sincosf(0.5f * (side == 0 ? er[k] + rp[k] : er[k] - rp[k]), &sn, &cs);
    reinterpret_cast<float2*>(out)[k] = make_float2(sn, cs);
  }
}

template <int NORM>
__global__ __launch_bounds__(256) void rank_dist_target_kernel(const float* __restrict__ Qv, const float* __restrict__ E,
                                                                const int32_t* __restrict__ triples, int64_t nq, int side, int N,
                                                                int K, int Kpad, float gamma, float modulus, float* __restrict__ tgt,
                                                                int32_t* __restrict__ counts, int zero_counts) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int t = triples[3 * q + (side == 0 ? 2 : 0)];
  tgt[q] = (unsigned)t < (unsigned)N ? dist_chain<NORM>(Qv + q * Kpad, E + (int64_t)t * K, K, gamma, modulus) : __builtin_nanf("");
  if (zero_counts) counts[q] = 0;
}

template <int NORM>
__global__ __launch_bounds__(256) void rank_dist_sweep_kernel(const float* __restrict__ Qv, const float* __restrict__ tgt,
                                                               int64_t nq, const float* __restrict__ Ec, int K, int Kpad,
                                                               int cand_rows, int tiles_per_block, float gamma, float modulus,
                                                               int32_t* __restrict__ counts) {
  constexpr int T = 64, BK = 32;  // 64 queries x 64 candidates per block, 32-deep k chunks
  __shared__ __attribute__((aligned(16))) float Qs[BK * T];
  __shared__ __attribute__((aligned(16))) float Es[BK * T];
  __shared__ int cnt_s[T];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int64_t q0 = (int64_t)blockIdx.x * T;
  const int nct = (cand_rows + T - 1) / T;
  const int ct_begin = blockIdx.y * tiles_per_block, ct_end = min(nct, ct_begin + tiles_per_block);
  float tg[4];
  int cnt[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int64_t row = q0 + ty * 4 + i; tg[i] = row < nq ? tgt[row] : __builtin_inff(); }
  if (tid < T) cnt_s[tid] = 0;
  const int srow = tid & 63, skq = tid >> 6;  // staging: lane <-> row, 4 k-quads per pass
  for (int ct = ct_begin; ct < ct_end; ++ct) {
    const int c0 = ct * T;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < Kpad; k0 += BK) {
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < BK / 16; ++pass) {
        const int kl = (pass * 4 + skq) * 4;  // k offset inside the chunk
        const int k = k0 + kl;
        float qv[4] = {0.f, 0.f, 0.f, 0.f}, ev[4] = {0.f, 0.f, 0.f, 0.f};
        if (q0 + srow < nq) {
          const float* src = Qv + (q0 + srow) * Kpad + k;  // zero padded to Kpad (multiple of 32 here)
#pragma unroll
          for (int u = 0; u < 4; ++u) qv[u] = src[u];
        }
        if (c0 + srow < cand_rows) {
          const float* src = Ec + (int64_t)(c0 + srow) * K + k;
#pragma unroll
          for (int u = 0; u < 4; ++u) if (k + u < K) ev[u] = src[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { Qs[(kl + u) * T + srow] = qv[u]; Es[(kl + u) * T + srow] = ev[u]; }
      }
      __syncthreads();
      if (NORM == 4) {
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {  // rows of (sin, cos) pairs: two k-rows of the tile per element
          const float4 qs4 = *reinterpret_cast<const float4*>(Qs + kk * T + ty * 4), qc4 = *reinterpret_cast<const float4*>(Qs + (kk + 1) * T + ty * 4);
          const float4 es4 = *reinterpret_cast<const float4*>(Es + kk * T + tx * 4), ec4 = *reinterpret_cast<const float4*>(Es + (kk + 1) * T + tx * 4);
          const float qs[4] = {qs4.x, qs4.y, qs4.z, qs4.w}, qc[4] = {qc4.x, qc4.y, qc4.z, qc4.w};
          const float es[4] = {es4.x, es4.y, es4.z, es4.w}, ec[4] = {ec4.x, ec4.y, ec4.z, ec4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = phase_pair_step(acc[i][j], qs[i], qc[i], es[j], ec[j]);
        }
      } else {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) {
          const float4 q4 = *reinterpret_cast<const float4*>(Qs + kk * T + ty * 4);
          const float4 e4 = *reinterpret_cast<const float4*>(Es + kk * T + tx * 4);
          const float qa[4] = {q4.x, q4.y, q4.z, q4.w}, ea[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = dist_step<NORM>(acc[i][j], qa[i], ea[j], modulus);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool valid = c0 + tx * 4 + j < cand_rows;
#pragma unroll
      for (int i = 0; i < 4; ++i) cnt[i] += (valid && dist_score<NORM>(acc[i][j], gamma, modulus) > tg[i]) ? 1 : 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) if (cnt[i]) atomicAdd(&cnt_s[ty * 4 + i], cnt[i]);
  __syncthreads();
  if (tid < T && q0 + tid < nq && cnt_s[tid]) atomicAdd(&counts[q0 + tid], cnt_s[tid]);
}

template <int NORM>
__global__ __launch_bounds__(256) void rank_dist_filter_kernel(const float* __restrict__ Qv, const float* __restrict__ tgt,
                                                                const float* __restrict__ Ec, int cand_offset, int cand_rows,
                                                                int K, int Kpad, float gamma, float modulus, const int32_t* __restrict__ pair_q,
                                                                const int32_t* __restrict__ filt_ids, int64_t npairs,
                                                                int32_t* __restrict__ counts, int q_lo = 0, int q_n = 0x7fffffff) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npairs) return;
  const int q = pair_q[e] - q_lo;
  if ((unsigned)q >= (unsigned)q_n) return;
  const int c = filt_ids[e] - cand_offset;
  if ((unsigned)c >= (unsigned)cand_rows) return;
  if (dist_chain<NORM>(Qv + (int64_t)q * Kpad, Ec + (int64_t)c * K, K, gamma, modulus) > tgt[q]) atomicSub(&counts[q], 1);
}

template <int NORM>
__global__ void rank_dist_scores_kernel(const float* __restrict__ Qv, const float* __restrict__ Ec, int K, int Kpad, float gamma, float modulus,
                                         int64_t nq, int cand_rows, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * cand_rows) return;
  const int64_t q = i / cand_rows;
  out[i] = dist_chain<NORM>(Qv + q * Kpad, Ec + (int64_t)(i - q * cand_rows) * K, K, gamma, modulus);
}

// Debug / test hook: the score matrix entry as the chain computes it.
__global__ void rank_chain_scores_kernel(const float* __restrict__ Qv, const float* __restrict__ Ec, int K, int Kpad,
                                          int64_t nq, int cand_rows, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * cand_rows) return;
  const int64_t q = i / cand_rows;
  const int c = (int)(i - q * cand_rows);
  out[i] = chain_dot(Qv + q * Kpad, Ec + (int64_t)c * K, K, Kpad);
}

// ---------------------------------------------------------------------------------------------------------------
int rank_blocks_target();  // api.hip (env KGE_RANK_BLOCKS)

static int pick_bk(int K) {
  // least padding among the chunk widths built; (BK+4)/4 must be odd
  const int cands[2] = {40, 32};
  int best = 32, waste = 1 << 30;
  for (int bk : cands) {
    const int w = (int)align_up(K, bk) - K;
    if (w < waste) { waste = w; best = bk; }
  }
  return best;
}

static int pick_bk(int K);
int rank_kpad(int K);
// rows the 256 x 128 kernel takes: float4-aligned widths of at least two k chunks (its first chunk opens the accumulators, its last one
// carries the epilogue); narrower rows (K <= 40: a sweep of no weight) take the generic kernel
static bool sweep5_rows(int K) { return K % 4 == 0 && rank_kpad(K) / pick_bk(K) >= 2; }

static int sweep_variant() {  // KGE_RANK_SWEEP=1 (A/B timing, tests) selects the generic (predicated) kernel for aligned tables too
  static const int v = [] { const char* e = getenv("KGE_RANK_SWEEP"); return e ? atoi(e) : 0; }();
  return v;
}

static int sweep5_slots() {  // workgroups of rank_sweep5_kernel: one per CU (137 KB of LDS, 512 threads)
  return kCUs;
}

static void launch_sweep(int BK, bool v4, dim3 grid, hipStream_t st, const float* Qv, const float* tgt, int64_t nq, const float* Ec,
                         int K, int Kpad, int cand_rows, int tpb, int32_t* counts, int epi = 0, const float* en_c = nullptr) {
  if (nq <= 0 || cand_rows <= 0) return;
  ProfScope prof_sweep(KGE_PROF_RANK_SWEEP, st);
#define KGE_SWEEP(KERNEL, BKV, FLAG) hipLaunchKernelGGL((KERNEL<BKV, FLAG>), grid, dim3(256), 0, st, Qv, tgt, nq, Ec, K, Kpad, cand_rows, tpb, counts)
  if (v4 && sweep5_rows(K) && sweep_variant() == 0 && nq < (int64_t)1 << 30) {
    const bool tail = Kpad != K;
    // 256 x 128 tiles, one 512-thread workgroup per CU.  The phase-aligned cut is taken when it leaves at most 1/32 of the CUs
    // without a workgroup (it cuts the fabric traffic, not the run time); otherwise equal contiguous cuts of the unit list.
    const int64_t nqt = cdiv(nq, kQT5), nct = cdiv(cand_rows, kTile);
    const int64_t units = nqt * nct;
    const int slots = sweep5_slots();
    int parts = 0;
    {
      if (nqt <= slots && units > slots) {
        const int64_t g = std::min<int64_t>(slots / nqt, nct);
        if (g >= 1 && nqt * g * 32 >= (int64_t)slots * 31) parts = (int)g;
      }
    }
    const dim3 pgrid((unsigned)(parts ? nqt * parts : std::min<int64_t>(units, slots)));
#define KGE_SWEEP5(BKV, FLAG)                                                                                                                        \
do {                                                                                                                                               \
  if (epi == 1) hipLaunchKernelGGL((rank_sweep5_kernel<BKV, FLAG, 1>), pgrid, dim3(512), 0, st, Qv, tgt, (int)nq, Ec, K, Kpad, cand_rows, counts, en_c, parts);      \
  else if (epi == 2) hipLaunchKernelGGL((rank_sweep5_kernel<BKV, FLAG, 2>), pgrid, dim3(512), 0, st, Qv, tgt, (int)nq, Ec, K, Kpad, cand_rows, counts, en_c, parts); \
  else hipLaunchKernelGGL((rank_sweep5_kernel<BKV, FLAG, 0>), pgrid, dim3(512), 0, st, Qv, tgt, (int)nq, Ec, K, Kpad, cand_rows, counts, en_c, parts);               \
} while (0)
    if (BK == 40) { if (tail) KGE_SWEEP5(40, true); else KGE_SWEEP5(40, false); }
    else { if (tail) KGE_SWEEP5(32, true); else KGE_SWEEP5(32, false); }
#undef KGE_SWEEP5
    return;
  } else {
    if (BK == 40) { if (v4) KGE_SWEEP(rank_sweep_kernel, 40, true); else KGE_SWEEP(rank_sweep_kernel, 40, false); }
    else { if (v4) KGE_SWEEP(rank_sweep_kernel, 32, true); else KGE_SWEEP(rank_sweep_kernel, 32, false); }
  }
#undef KGE_SWEEP
}

#ifdef KGE_SWEEP5_STAMP
extern "C" int kge_debug_sweep5_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep5_stamps), sizeof(unsigned long long) * 256 * kStampUnits) == hipSuccess ? 0 : -1;
}
#endif

int rank_kpad(int K) { return (int)align_up(K, pick_bk(K)); }
static int rank_kpad_model(const kge_model_desc* md) { return md->model == KGE_TRANSE ? (int)align_up(md->d_e, 32) : rank_kpad(md->d_e); }

int64_t rank_workspace_bytes(const kge_model_desc* md, int64_t nq) {
  // TransE-L2 on the matrix cores additionally keeps the squared norm of every table row (rank_sumsq_kernel)
  int64_t en = md->model == KGE_TRANSE && md->norm == 2 ? align_up(md->num_entities, 64) : 0;
  int64_t qrow = rank_kpad_model(md);
  if (md->model == KGE_TRANSE && md->norm == 3) {  // pRotatE: (sin, cos) pair rows of the queries and of the whole table
    qrow = align_up(2 * (int64_t)md->d_e, 32);
    en = align_up(2 * (int64_t)md->d_e * md->num_entities, 64);
  }
  return (int64_t)sizeof(float) * (nq * qrow + align_up(nq, 64) + en) + 512;
}

static bool rank_phase_pairs() {  // KGE_RANK_PHASE=0: pRotatE ranks with one sinf per element (read per call: tests flip it)
  const char* e = getenv("KGE_RANK_PHASE");
  return !(e && atoi(e) == 0);
}
static bool rank_l2_mfma() {  // KGE_RANK_L2=0: TransE-L2 ranks through the VALU distance sweep (read per call: tests flip it)
  const char* e = getenv("KGE_RANK_L2");
  return !(e && atoi(e) == 0);
}

static void launch_build_q(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t nq, int side,
                           int Kpad, float* Qv, uint32_t* err, hipStream_t st) {
  if (md->model == KGE_HOLE) { (void)hole_build_q(md, E, Rel, triples, nq, side, Kpad, Qv, err, st); return; }
  const int grid = (int)cdiv(nq * 64, 256);
  const int K = md->d_e;
  if (md->model == KGE_DISTMULT)
    hipLaunchKernelGGL((rank_build_q_kernel<KGE_DISTMULT>), dim3(grid), dim3(256), 0, st, E, Rel, triples, nq, side, md->num_entities, md->num_relations, K, Kpad, Qv, err, 0.f);
  else if (md->model == KGE_TRANSE)
    hipLaunchKernelGGL((rank_build_q_kernel<KGE_TRANSE>), dim3(grid), dim3(256), 0, st, E, Rel, triples, nq, side, md->num_entities, md->num_relations, K, Kpad, Qv, err, 0.f);
  else if (md->model == KGE_ROTATE)
    hipLaunchKernelGGL((rank_build_q_kernel<KGE_ROTATE>), dim3(grid), dim3(256), 0, st, E, Rel, triples, nq, side, md->num_entities, md->num_relations, K, Kpad, Qv, err, md->gamma);
  else
    hipLaunchKernelGGL((rank_build_q_kernel<KGE_COMPLEX>), dim3(grid), dim3(256), 0, st, E, Rel, triples, nq, side, md->num_entities, md->num_relations, K, Kpad, Qv, err, 0.f);
}

template <int NORM>
static void launch_dist(const kge_model_desc* md, const float* E, const float* Ec, int cand_offset, int cand_rows, const int32_t* triples,
                        int64_t nq, int side, const int32_t* pair_q, const int32_t* filt_ids, int64_t npairs, int32_t* counts,
                        int accumulate, const float* Qv, float* tgt, int Kpad, hipStream_t st, int q_lo = 0) {
  const int K = md->d_e;
  hipLaunchKernelGGL((rank_dist_target_kernel<NORM>), dim3((int)cdiv(nq, 256)), dim3(256), 0, st, Qv, E, triples, nq, side, md->num_entities, K, Kpad, md->gamma, md->modulus, tgt, counts, accumulate ? 0 : 1);
  const int nqt = (int)cdiv(nq, 64), nct = (int)cdiv(cand_rows, 64);
  int splits = std::max(1, std::min((int)cdiv(rank_blocks_target(), nqt), nct));
  const int tpb = (int)cdiv(nct, splits);
  splits = (int)cdiv(nct, tpb);
  hipLaunchKernelGGL((rank_dist_sweep_kernel<NORM>), dim3(nqt, splits), dim3(256), 0, st, Qv, tgt, nq, Ec, K, Kpad, cand_rows, tpb, md->gamma, md->modulus, counts);
  if (npairs > 0)
    hipLaunchKernelGGL((rank_dist_filter_kernel<NORM>), dim3((int)cdiv(npairs, 256)), dim3(256), 0, st, Qv, tgt, Ec, cand_offset, cand_rows, K, Kpad, md->gamma, md->modulus, pair_q, filt_ids, npairs, counts, q_lo, (int)nq);
}

// side 0 / 1: nt tail / head queries.  side 2: both sides of the nt test triples in ONE sweep -- queries [0, nt) are the tail queries,
// [nt, 2 nt) the head queries (counts, target scores and the filter list's query numbers follow that numbering): the candidate table
// is swept once for twice the queries (fewer, fuller query tiles; one launch instead of two).  Where the model's sweep is not one of
// the matrix-core sweeps the two sides run one after the other inside this call.  q_lo: the number of this call's first query in the
// filter list (inner calls of side 2).
static int rank_counts_impl(const kge_model_desc* md, const float* E, const float* Rel, const float* Ec, int cand_offset, int cand_rows,
                            const int32_t* triples, int64_t nt, int side, const int32_t* pair_q, const int32_t* filt_ids, int64_t npairs,
                            int32_t* counts, int accumulate, float* tgt_out, void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st, int q_lo) {
  if (md->model == KGE_HOLE && md->d_e > 1024) return fail(KGE_EUNSUPPORTED, "kge_rank_counts: HolE rows wider than 1024 floats are not built");
  const int K = md->d_e, BK = pick_bk(K);
  int Kpad = rank_kpad_model(md);
  const int nside = side == 2 ? 2 : 1;
  const int64_t nq = nside * nt;  // queries of this call
  if (wbytes < rank_workspace_bytes(md, nq)) return fail(KGE_EWORKSPACE, "kge_rank_counts: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)rank_workspace_bytes(md, nq));
  uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  float* Qv = reinterpret_cast<float*>(p);
  float* tgt = tgt_out ? tgt_out : Qv + nq * Kpad;
  const bool cand_in_table0 = Ec == E + (int64_t)cand_offset * K && cand_offset + cand_rows <= md->num_entities;
  const bool mfma_ok = sweep5_rows(K) && (((uintptr_t)E | (uintptr_t)Ec) % 16 == 0) && nq < (int64_t)1 << 30 && sweep_variant() == 0;
  const bool merged = md->model == KGE_TRANSE ? (md->norm == 2 && !tgt_out && rank_l2_mfma() && mfma_ok && cand_in_table0) : mfma_ok;
  if (side == 2 && !merged) {
    for (int h = 0; h < 2; ++h)
      if (int rc = rank_counts_impl(md, E, Rel, Ec, cand_offset, cand_rows, triples, nt, h, pair_q, filt_ids, npairs, counts + h * nt, accumulate,
                                    tgt_out ? tgt_out + h * nt : nullptr, workspace, wbytes, err, st, q_lo + (int)(h * nt))) return rc;
    return KGE_OK;
  }
  const int qn = (int)nq;  // the filter kernels' query window: [q_lo, q_lo + qn)
  // TransE-L2 on the matrix cores (target scores are not produced on this path: a caller asking for them gets the distance sweep)
  // (the per-row tables of the two forms below -- squared norms, (sin, cos) pairs -- are built from E and indexed by global id, so the
  // candidates must be rows of E itself; a row shard held apart from the query table takes the direct distance sweep)
  const bool cand_in_table = cand_in_table0;
  if (md->model == KGE_TRANSE && md->norm == 2 && !tgt_out && rank_l2_mfma() && sweep5_rows(K) && (((uintptr_t)E | (uintptr_t)Ec) % 16 == 0) &&
      nq < (int64_t)1 << 30 && sweep_variant() == 0 && cand_in_table) {
    float* en = tgt + align_up(nq, 64);  // behind the target values (the query block is sized for align_up(K, 32) >= rank_kpad(K) floats per row)
    Kpad = rank_kpad(K);
    hipLaunchKernelGGL(rank_sumsq_kernel, dim3((int)cdiv((int64_t)md->num_entities * 16, 256)), dim3(256), 0, st, E, md->num_entities, K, en);
    for (int h = 0; h < nside; ++h) {
      const int sd = side == 2 ? h : side;
      float* Qh = Qv + h * nt * Kpad;
      launch_build_q(md, E, Rel, triples, nt, sd, Kpad, Qh, err, st);
      if (BK == 40) hipLaunchKernelGGL((rank_target4_kernel<40, true>), dim3((int)cdiv(nt, 64)), dim3(64), 0, st, Qh, E, triples, (int)nt, sd, md->num_entities, K, Kpad, tgt + h * nt, counts + h * nt, accumulate ? 0 : 1, en);
      else hipLaunchKernelGGL((rank_target4_kernel<32, true>), dim3((int)cdiv(nt, 64)), dim3(64), 0, st, Qh, E, triples, (int)nt, sd, md->num_entities, K, Kpad, tgt + h * nt, counts + h * nt, accumulate ? 0 : 1, en);
    }
    const bool less = md->gamma > 0.f;
    launch_sweep(BK, true, dim3(1), st, Qv, tgt, nq, Ec, K, Kpad, cand_rows, 1, counts, less ? 1 : 2, en + cand_offset);
    if (npairs > 0) {
      if (less) hipLaunchKernelGGL((rank_filter_l2_kernel<true>), dim3((int)cdiv(npairs, 256)), dim3(256), 0, st, Qv, tgt, Ec, en + cand_offset, cand_offset, cand_rows, K, Kpad, pair_q, filt_ids, npairs, counts, q_lo, qn);
      else hipLaunchKernelGGL((rank_filter_l2_kernel<false>), dim3((int)cdiv(npairs, 256)), dim3(256), 0, st, Qv, tgt, Ec, en + cand_offset, cand_offset, cand_rows, K, Kpad, pair_q, filt_ids, npairs, counts, q_lo, qn);
    }
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  if (md->model == KGE_TRANSE && md->norm == 3 && rank_phase_pairs() && cand_in_table) {
    // pRotatE on (sin, cos) pair rows (NORM 4 of the distance kernels): tables once per call, then 3 VALU operations per element
    const int K2 = 2 * K, Kpad2 = (int)align_up(K2, 32);
    float* tg = tgt_out ? tgt_out : Qv + nq * Kpad2;
    float* EP = Qv + nq * Kpad2 + align_up(nq, 64);
    const int64_t ne = (int64_t)md->num_entities * K;
    hipLaunchKernelGGL(rank_phase_table_kernel, dim3((unsigned)cdiv(ne, 256)), dim3(256), 0, st, E, ne, EP);
    hipLaunchKernelGGL(rank_build_q_phase_kernel, dim3((unsigned)cdiv(nq * 64, 256)), dim3(256), 0, st, E, Rel, triples, nq, side, md->num_entities,
                       md->num_relations, K, Kpad2, Qv, err);
    kge_model_desc m2 = *md;
    m2.d_e = m2.d_r = K2;
    launch_dist<4>(&m2, EP, EP + (int64_t)cand_offset * K2, cand_offset, cand_rows, triples, nq, side, pair_q, filt_ids, npairs, counts, accumulate, Qv, tg,
                   Kpad2, st, q_lo);
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  if (md->model != KGE_TRANSE) {
    for (int h = 0; h < nside; ++h) launch_build_q(md, E, Rel, triples, nt, side == 2 ? h : side, Kpad, Qv + h * nt * Kpad, err, st);
  } else {
    launch_build_q(md, E, Rel, triples, nq, side, Kpad, Qv, err, st);
  }
  if (md->model == KGE_TRANSE) {
    if (md->norm == 1) launch_dist<1>(md, E, Ec, cand_offset, cand_rows, triples, nq, side, pair_q, filt_ids, npairs, counts, accumulate, Qv, tgt, Kpad, st, q_lo);
    else if (md->norm == 3) launch_dist<3>(md, E, Ec, cand_offset, cand_rows, triples, nq, side, pair_q, filt_ids, npairs, counts, accumulate, Qv, tgt, Kpad, st, q_lo);
    else if (md->norm == 5) launch_dist<5>(md, E, Ec, cand_offset, cand_rows, triples, nq, side, pair_q, filt_ids, npairs, counts, accumulate, Qv, tgt, Kpad, st, q_lo);
    else launch_dist<2>(md, E, Ec, cand_offset, cand_rows, triples, nq, side, pair_q, filt_ids, npairs, counts, accumulate, Qv, tgt, Kpad, st, q_lo);
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  if ((K % 4 == 0) && ((uintptr_t)E % 16 == 0) && nq < (int64_t)1 << 30 && sweep_variant() == 0) {
    for (int h = 0; h < nside; ++h) {
      const int sd = side == 2 ? h : side;
      if (BK == 40) hipLaunchKernelGGL((rank_target4_kernel<40>), dim3((int)cdiv(nt, 64)), dim3(64), 0, st, Qv + h * nt * Kpad, E, triples, (int)nt, sd, md->num_entities, K, Kpad, tgt + h * nt, counts + h * nt, accumulate ? 0 : 1);
      else hipLaunchKernelGGL((rank_target4_kernel<32>), dim3((int)cdiv(nt, 64)), dim3(64), 0, st, Qv + h * nt * Kpad, E, triples, (int)nt, sd, md->num_entities, K, Kpad, tgt + h * nt, counts + h * nt, accumulate ? 0 : 1);
    }
  } else {  // (never with side 2: that needs the conditions of this branch's `if`)
    hipLaunchKernelGGL(rank_target_kernel, dim3((int)cdiv(nq, 256)), dim3(256), 0, st, Qv, E, triples, nq, side, md->num_entities, K, Kpad, tgt, counts, accumulate ? 0 : 1);
  }
  {
    const int nqt = (int)cdiv(nq, kTile), nct = (int)cdiv(cand_rows, kTile);
    int splits = (int)cdiv(rank_blocks_target(), nqt);  // >= ~16 blocks per CU so the last, partially filled round is short
    splits = std::max(1, std::min(splits, nct));
    const int tpb = std::min(60, (int)cdiv(nct, splits));  // packed 8-bit epilogue counters: <= 60 tiles per block
    splits = (int)cdiv(nct, tpb);
    const bool v4 = (K % 4 == 0) && ((uintptr_t)Ec % 16 == 0);
    dim3 grid(nqt, splits);
    launch_sweep(BK, v4, grid, st, Qv, tgt, nq, Ec, K, Kpad, cand_rows, tpb, counts);
  }
  if (npairs > 0)
    hipLaunchKernelGGL(rank_filter_kernel, dim3((int)cdiv(npairs, 256)), dim3(256), 0, st, Qv, tgt, Ec, cand_offset, cand_rows, K, Kpad, pair_q, filt_ids, npairs, counts, q_lo, qn);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int rank_counts(const kge_model_desc* md, const float* E, const float* Rel, const float* Ec, int cand_offset, int cand_rows,
                const int32_t* triples, int64_t nq, int side, const int32_t* pair_q, const int32_t* filt_ids, int64_t npairs,
                int32_t* counts, int accumulate, float* tgt_out, void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st) {
  ProfScope prof(KGE_PROF_RANK, st);
  return rank_counts_impl(md, E, Rel, Ec, cand_offset, cand_rows, triples, nq, side, pair_q, filt_ids, npairs, counts, accumulate, tgt_out, workspace,
                          wbytes, err, st, 0);
}

// ---- generic entry: queries given as vectors (ConvE's hidden vectors; head queries against a per-relation candidate matrix)
__global__ void rank_pad_q_kernel(const float* __restrict__ Q, int64_t nq, int K, int Kpad, float* __restrict__ Qv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * Kpad) return;
  const int64_t q = i / Kpad;
  const int k = (int)(i - q * Kpad);
  Qv[i] = k < K ? Q[q * K + k] : 0.f;
}
__global__ __launch_bounds__(256) void rank_target_ids_kernel(const float* __restrict__ Qv, const float* __restrict__ Cand, int cand_offset,
                                                               int cand_rows, const int32_t* __restrict__ target_ids, int64_t nq, int K,
                                                               int Kpad, float* __restrict__ tgt, int32_t* __restrict__ counts, int zero_counts) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int t = target_ids[q] - cand_offset;
  tgt[q] = (unsigned)t < (unsigned)cand_rows ? chain_dot(Qv + q * Kpad, Cand + (int64_t)t * K, K, Kpad) : __builtin_nanf("");
  if (zero_counts) counts[q] = 0;
}

int64_t rank_sweep_q_workspace_bytes(int64_t nq, int K) { return (int64_t)sizeof(float) * (nq * rank_kpad(K) + align_up(nq, 64)) + 512; }

int rank_sweep_q(const float* Q, int64_t nq, int K, const float* Ec, int cand_offset, int cand_rows, const int32_t* target_ids,
                 const int32_t* pair_q, const int32_t* filt_ids, int64_t npairs, int32_t* counts, int accumulate, void* workspace,
                 int64_t wbytes, hipStream_t st) {
  if (wbytes < rank_sweep_q_workspace_bytes(nq, K)) return fail(KGE_EWORKSPACE, "kge_rank_sweep_q: workspace too small");
  const int BK = pick_bk(K), Kpad = (int)align_up(K, BK);
  uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  float* Qv = reinterpret_cast<float*>(p);
  float* tgt = Qv + nq * Kpad;
  ProfScope prof(KGE_PROF_RANK, st);
  hipLaunchKernelGGL(rank_pad_q_kernel, dim3((int)cdiv(nq * Kpad, 256)), dim3(256), 0, st, Q, nq, K, Kpad, Qv);
  hipLaunchKernelGGL(rank_target_ids_kernel, dim3((int)cdiv(nq, 256)), dim3(256), 0, st, Qv, Ec, cand_offset, cand_rows, target_ids, nq, K, Kpad, tgt,
                     counts, accumulate ? 0 : 1);
  {
    const int nqt = (int)cdiv(nq, kTile), nct = (int)cdiv(cand_rows, kTile);
    int splits = (int)cdiv(rank_blocks_target(), nqt);
    splits = std::max(1, std::min(splits, nct));
    const int tpb = std::min(60, (int)cdiv(nct, splits));
    splits = (int)cdiv(nct, tpb);
    const bool v4 = (K % 4 == 0) && ((uintptr_t)Ec % 16 == 0);
    dim3 grid(nqt, splits);
    launch_sweep(BK, v4, grid, st, Qv, tgt, nq, Ec, K, Kpad, cand_rows, tpb, counts);
  }
  if (npairs > 0)
    hipLaunchKernelGGL(rank_filter_kernel, dim3((int)cdiv(npairs, 256)), dim3(256), 0, st, Qv, tgt, Ec, cand_offset, cand_rows, K, Kpad, pair_q, filt_ids, npairs, counts);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int rank_chain_scores(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t nq, int side,
                      float* out, void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st) {
  if (md->model == KGE_HOLE && md->d_e > 1024) return fail(KGE_EUNSUPPORTED, "kge_rank_chain_scores: HolE rows wider than 1024 floats are not built");
  const int K = md->d_e, Kpad = rank_kpad_model(md);
  if (wbytes < rank_workspace_bytes(md, nq)) return fail(KGE_EWORKSPACE, "kge_rank_chain_scores: workspace too small");
  uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  float* Qv = reinterpret_cast<float*>(p);
  launch_build_q(md, E, Rel, triples, nq, side, Kpad, Qv, err, st);
  const int64_t tot = nq * md->num_entities;
  if (md->model == KGE_TRANSE) {
    if (md->norm == 1) hipLaunchKernelGGL((rank_dist_scores_kernel<1>), dim3((int)cdiv(tot, 256)), dim3(256), 0, st, Qv, E, K, Kpad, md->gamma, md->modulus, nq, md->num_entities, out);
    else if (md->norm == 3) hipLaunchKernelGGL((rank_dist_scores_kernel<3>), dim3((int)cdiv(tot, 256)), dim3(256), 0, st, Qv, E, K, Kpad, md->gamma, md->modulus, nq, md->num_entities, out);
    else if (md->norm == 5) hipLaunchKernelGGL((rank_dist_scores_kernel<5>), dim3((int)cdiv(tot, 256)), dim3(256), 0, st, Qv, E, K, Kpad, md->gamma, md->modulus, nq, md->num_entities, out);
    else hipLaunchKernelGGL((rank_dist_scores_kernel<2>), dim3((int)cdiv(tot, 256)), dim3(256), 0, st, Qv, E, K, Kpad, md->gamma, md->modulus, nq, md->num_entities, out);
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  hipLaunchKernelGGL(rank_chain_scores_kernel, dim3((int)cdiv(tot, 256)), dim3(256), 0, st, Qv, E, K, Kpad, nq, md->num_entities, out);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
