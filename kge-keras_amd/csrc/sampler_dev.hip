// Device-side, bit-exact batch builder: the draws of generate_negative + oversample_data (KGEkeras/utils.py:114-147,
// driven as in tests.py:55-61) produced on the GPU, so that a training batch never leaves HBM.
//
// Both of the reference's generators are MT19937.  Its state transition is sequential over 624-word blocks, but inside
// one block word i depends on words i, i+1 and i+397 (mod 624), so 227 words can be produced at a time: one workgroup
// regenerates a block in three barrier-separated phases (0..226 | 227..453 | 454..623).  Everything after that is
// embarrassingly parallel:
//
//   mt_generate_kernel   2 workgroups (NumPy stream, Python stream): K blocks of raw state words each  -> U_np, U_py
//   np_select_kernel     temper + mask + accept (NumPy's masked rejection, utils.py:118-119), per-tile accept counts published
//                        and summed in the same pass; the first m accepted draws -> fs, the next m -> fo; position of the
//                        2m-th -> words consumed
//   py_assemble_kernel   choices: idx = floor(((a>>5)*2^26 + (b>>6)) / 2^53 * n) from word pairs (utils.py:136),
//                        then pos_out[i] = batch[idxA[i]], neg_out[i] = (fs[j], batch[j/ns].p, fo[j]), j = idxB[i]
//   mt_advance_kernel    new generator states (the block the consumption stopped in + position inside it)
//
// The NumPy stream consumes a data-dependent number of words; the raw buffer is sized with a margin of tens of
// standard deviations and err_flag bit 1 is raised if it ever ran short (never observed).  There is no fallback: the batch
// drawn is then NOT the reference's, and the callers (harness.DeviceDataGenerator.release_global, pipeline.TrainPipeline.finish,
// bench.py) read the flag and raise.
#include <cstring>

#include "kge_internal.hpp"

namespace kge {

struct MTDev {
  uint32_t mt[624];
  int32_t pos;
  int32_t pad;
};
static_assert(sizeof(MTDev) == KGE_MT_STATE_BYTES, "state blob size");

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
  const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
  return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// blockIdx.x == 0: NumPy stream (K0 blocks), 1: Python stream (K1 blocks).  U holds RAW state words (untempered), so
// any block of U is itself a valid generator state.
// The barrier is a raw s_barrier behind an LDS-only wait: __syncthreads() would also drain the global stores of the
// previous block (vmcnt(0)) on the critical path.
#define KGE_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// Regenerates blocks [b_begin, b_end) of U from the state block held in buf[0] (256 threads, all must call).
// Thread tid owns words tid, 227+tid and 454+tid of a block.  The phase-2 / phase-3 inputs new[i-227] are then the
// thread's OWN phase-1 / phase-2 results, so a whole block is computed from one round of LDS reads of the previous
// block: one barrier per block instead of three.  (new[623] needs new[0]: thread 169 recomputes it.)
__device__ __forceinline__ void mt_regen_blocks(uint32_t (*buf)[624], uint32_t* __restrict__ U, int b_begin, int b_end) {
  const int tid = threadIdx.x;
  int cur = 0;
  for (int b = b_begin; b < b_end; ++b) {
    const uint32_t* o = buf[cur];
    uint32_t* n = buf[cur ^ 1];
    uint32_t* dst = U + (int64_t)b * 624;
    if (tid < 227) {
      const uint32_t a0 = o[tid], a1 = o[tid + 1], a397 = o[tid + 397];
      const uint32_t b0 = o[227 + tid], b1 = o[228 + tid];
      const uint32_t v0 = a397 ^ mt_twist(a0, a1);
      const uint32_t v1 = v0 ^ mt_twist(b0, b1);
      n[tid] = v0;
      n[227 + tid] = v1;
      dst[tid] = v0;
      dst[227 + tid] = v1;
      if (tid < 169) {
        const uint32_t v2 = v1 ^ mt_twist(o[454 + tid], o[455 + tid]);
        n[454 + tid] = v2;
        dst[454 + tid] = v2;
      } else if (tid == 169) {
        const uint32_t n0 = o[397] ^ mt_twist(o[0], o[1]);
        const uint32_t v2 = v1 ^ mt_twist(o[623], n0);  // v1 of thread 169 is new[396]
        n[623] = v2;
        dst[623] = v2;
      }
    }
    KGE_LDS_BARRIER();
    cur ^= 1;
  }
}

// blockIdx.x == 0: NumPy stream (K0 blocks), 1: Python stream (K1 blocks), generated sequentially from the state.
constexpr int kSeg = 16;
__global__ __launch_bounds__(256) void mt_generate_kernel(const MTDev* __restrict__ st_np, const MTDev* __restrict__ st_py,
                                                           uint32_t* __restrict__ U_np, int K0, uint32_t* __restrict__ U_py, int K1,
                                                           uint32_t* __restrict__ Jst_zero) {
  __shared__ uint32_t buf[2][624];
  const MTDev* st = blockIdx.x == 0 ? st_np : st_py;
  __builtin_amdgcn_s_setprio(3);  // a latency chain of a few workgroups beside chip-filling gather kernels: win the issue arbitration
  for (int i = threadIdx.x; i < 624; i += 256) buf[0][i] = st->mt[i];
  (void)Jst_zero;
  __syncthreads();
  mt_regen_blocks(buf, blockIdx.x == 0 ? U_np : U_py, 0, blockIdx.x == 0 ? K0 : K1);
}

// ---- segmented generation: GF(2) jump-ahead (mtjump.cpp) gives the state at the start of every segment as a XOR of
// windows of the first 33 blocks of the raw stream, so kSeg workgroups per stream regenerate in parallel.
constexpr int kJumpBlocks = 33;   // 624 * 33 >= 19937 + 623 raw words after the state block
constexpr int kPolyWords = 312;   // uint64 words of one jump polynomial

// Jump-ahead as a window XOR: state_J[j] = XOR_{k : g_J[k]} W[k + j], W = the state block followed by the raw blocks behind it.
// Lane l of a wave stands for COEFFICIENT k0 + l of the current polynomial word, not for an output word: the 64 coefficient bits of
// the word become the execution mask of the wave (`if (on)`), under which each of the wave's 16 output words j takes one
// conflict-free LDS read W[k0 + l + j] and one XOR -- no per-bit scalar test at all (a lane per output word with the bit tested in
// scalar registers costs two scalar instructions per LDS read, and the one scalar unit of a CU serves four SIMDs: 28.6 us for the 30
// jumps of a batch).  The 64 lanes' partial words meet in an XOR butterfly at the end.  Work unit of a workgroup: (polynomial,
// 64 output words, one of kJumpSplit coefficient slices); partial results go to Jpart[z][s][slice][j] (no atomics, nothing to zero) and
// the segment kernel XORs the slices when it loads its start state.
constexpr int kJumpSplit = 4;
constexpr int kJRow = 640;   // 624 padded
constexpr int kJTile = 16;   // output words per wave
template <typename LoadW>
__device__ __forceinline__ void jump_core(LoadW loadw, const uint64_t* __restrict__ polys_z, uint32_t* __restrict__ Jpart_z) {
  constexpr int WPS = kPolyWords / kJumpSplit;                     // polynomial words per slice (78)
  static_assert(kPolyWords % kJumpSplit == 0, "slices must divide the polynomial");
  constexpr int WIN = WPS * 64 + 4 * kJTile + 64;                  // raw words this workgroup touches
  __shared__ uint32_t win[WIN];
  __shared__ uint64_t pw[WPS];                                     // this slice's polynomial words (read back as LDS broadcasts)
  const int jt = blockIdx.x % 10, slice = blockIdx.x / 10, s = blockIdx.y;  // s: polynomial 0 .. kSeg-2 (segment s + 1)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = jt * 64 + wave * kJTile;                          // first output word of this wave
  const int lo = slice * WPS * 64 + jt * 64;                       // first raw word of the window
  for (int i = threadIdx.x; i < WIN; i += 256) {
    const int idx = lo + i;
    win[i] = idx < 624 * (kJumpBlocks + 1) ? loadw(idx) : 0u;
  }
  if (threadIdx.x < WPS) pw[threadIdx.x] = polys_z[(int64_t)s * kPolyWords + slice * WPS + threadIdx.x];
  __syncthreads();
  uint32_t acc[kJTile];
#pragma unroll
  for (int u = 0; u < kJTile; ++u) acc[u] = 0u;
  const uint32_t* src = win + wave * kJTile + lane;
#pragma unroll 2
  for (int w = 0; w < WPS; ++w) {
    const uint64_t g = pw[w];                                      // wave-uniform
    const bool on = (g >> lane) & 1ull;
    if (on) {
#pragma unroll
      for (int u = 0; u < kJTile; ++u) acc[u] ^= src[w * 64 + u];
    }
  }
#pragma unroll
  for (int u = 0; u < kJTile; ++u) {
    uint32_t v = acc[u];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= (uint32_t)__shfl_xor((int)v, d, 64);
    acc[u] = v;
  }
  if (lane < kJTile && j0 + lane < 624) {
    uint32_t v = 0u;
#pragma unroll
    for (int u = 0; u < kJTile; ++u) v = lane == u ? acc[u] : v;
    Jpart_z[((int64_t)(s + 1) * kJumpSplit + slice) * kJRow + j0 + lane] = v;
  }
}
// start state of segment s (>= 1) of generator z: XOR of the slices' partial results
__device__ __forceinline__ uint32_t jump_state_word(const uint32_t* __restrict__ Jpart, int z, int s, int i) {
  const uint32_t* p = Jpart + (((int64_t)z * kSeg + s) * kJumpSplit) * kJRow + i;
  uint32_t v = 0u;
#pragma unroll 8
  for (int sl = 0; sl < kJumpSplit; ++sl) v ^= p[(int64_t)sl * kJRow];
  return v;
}

// grid (10 j-tiles x kJumpSplit k-slices, kSeg-1 polynomials, 2 generators)
__global__ __launch_bounds__(256) void mt_jump_kernel(const MTDev* __restrict__ st_np, const MTDev* __restrict__ st_py,
                                                       const uint32_t* __restrict__ U_np, const uint32_t* __restrict__ U_py,
                                                       const uint64_t* __restrict__ polys, uint32_t* __restrict__ Jpart, int seg_np,
                                                       int seg_py) {
  const int z = blockIdx.z;
  if ((z == 0 ? seg_np : seg_py) == 0) return;  // this stream is generated sequentially
  const MTDev* st = z == 0 ? st_np : st_py;
  const uint32_t* U = z == 0 ? U_np : U_py;
  jump_core([&](int idx) { return idx < 624 ? st->mt[idx] : U[idx - 624]; }, polys + (int64_t)z * (kSeg - 1) * kPolyWords,
            Jpart + (int64_t)z * kSeg * kJumpSplit * kJRow);
}

// grid (kSeg, 2): segment s of stream z regenerates blocks [s*segb, (s+1)*segb) (segment 0 starts after the 33 blocks
// the sequential pre-pass already produced).
__global__ __launch_bounds__(256) void mt_generate_seg_kernel(uint32_t* __restrict__ U_np, int K_np, int segb_np,
                                                               uint32_t* __restrict__ U_py, int K_py, int segb_py,
                                                               const uint32_t* __restrict__ Jst) {
  __shared__ uint32_t buf[2][624];
  const int z = blockIdx.y, s = blockIdx.x;
  uint32_t* U = z == 0 ? U_np : U_py;
  const int K = z == 0 ? K_np : K_py, segb = z == 0 ? segb_np : segb_py;
  if (segb == 0) return;
  int b0 = s * segb, b1 = min(K, (s + 1) * segb);
  if (s == 0) b0 = kJumpBlocks;
  if (b0 >= b1) return;
  __builtin_amdgcn_s_setprio(3);
  for (int i = threadIdx.x; i < 624; i += 256) buf[0][i] = s == 0 ? U[(int64_t)(kJumpBlocks - 1) * 624 + i] : jump_state_word(Jst, z, s, i);
  __syncthreads();
  mt_regen_blocks(buf, U, b0, b1);
}
#undef KGE_LDS_BARRIER

// word `i` of the stream that starts at the generator's current position: tail of the state, then U
__device__ __forceinline__ uint32_t stream_word(const MTDev* st, const uint32_t* U, int p0, int64_t i) {
  const int tail = 624 - p0;
  return mt_temper(i < tail ? st->mt[p0 + i] : U[i - tail]);
}

constexpr int kTileW = 1024;  // stream words per tile (256 threads x 4)

// NumPy's masked rejection (utils.py:118-119) in ONE pass: every tile of 1024 stream words tempers + masks + tests its words,
// publishes its accept count tagged with the batch epoch, sums the counts of all tiles before it (they publish as soon as
// they have counted and wait for nothing, and workgroups are dispatched in order, so the wait always ends) and scatters its
// accepted draws: the first m -> fs, the next m -> fo; the position of the 2m-th gives the words consumed.
// tile_agg[b] = epoch << 32 | count.  The epoch lives in the (zero-initialised) workspace and is advanced by mt_advance_kernel.
__global__ __launch_bounds__(256) void np_select_kernel(const MTDev* __restrict__ st, const uint32_t* __restrict__ U, int64_t L_u,
                                                         uint32_t rng, uint32_t mask, unsigned long long* tile_agg,
                                                         const uint32_t* __restrict__ epoch_ptr, int64_t m, int32_t* __restrict__ fs,
                                                         int32_t* __restrict__ fo, int64_t* __restrict__ consumed,
                                                         int32_t* __restrict__ zero_p, int64_t zero_n) {
  __shared__ int wsum[4];
  __shared__ long long pre_s[4];
  // (the pick counters py_assemble_kernel adds into are cleared here, on the way)
  for (int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; z < zero_n; z += (int64_t)gridDim.x * blockDim.x) zero_p[z] = 0;
  const unsigned long long ep = (unsigned long long)(epoch_ptr[0] + 1u);
  const int p0 = st->pos;
  const int64_t L = (624 - p0) + L_u;
  const int64_t base = (int64_t)blockIdx.x * kTileW + threadIdx.x * 4;
  uint32_t v[4];
  int a[4], c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = base + j < L ? (stream_word(st, U, p0, base + j) & mask) : 0xFFFFFFFFu;
    a[j] = (base + j < L && v[j] <= rng) ? 1 : 0;
    c += a[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = c;
  for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(inc, d, 64); if (lane >= d) inc += up; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(tile_agg + blockIdx.x, ep << 32 | (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // counts of the tiles in front of this one
  long long before = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) {
    unsigned long long s;
    do { s = __hip_atomic_load(tile_agg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((s >> 32) != ep);
    before += (long long)(s & 0xFFFFFFFFull);
  }
  for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) pre_s[wave] = before;
  __syncthreads();
  int64_t r = pre_s[0] + pre_s[1] + pre_s[2] + pre_s[3] + inc - c;
  for (int w = 0; w < wave; ++w) r += wsum[w];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!a[j]) continue;
    if (r < m) fs[r] = (int32_t)v[j];
    else if (r < 2 * m) fo[r - m] = (int32_t)v[j];
    if (r == 2 * m - 1) { consumed[0] = base + j + 1; consumed[1] = (int64_t)ep; }  // tagged: no reset between batches
    ++r;
  }
}

// CPython random.random() from two consecutive outputs; choices index = floor(random() * n)
__device__ __forceinline__ int64_t py_choice(const MTDev* st, const uint32_t* U, int p0, int64_t i, double n) {
  const uint32_t a = stream_word(st, U, p0, 2 * i) >> 5, b = stream_word(st, U, p0, 2 * i + 1) >> 6;
  const double r = ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
  return (int64_t)floor(r * n);
}

// The positives of the batch: a (B,3) array, or rows index[starts[*cursor] * B + i] of the training triples (the slice
// DataGenerator.__getitem__ takes of its epoch permutation, tests.py:43-47,55-56) read in place.
struct BatchSrc {
  const int32_t* batch;
  const int32_t* kg;
  const int64_t* index;
  const int64_t* starts;
  const int64_t* cursor;
};
__device__ __forceinline__ const int32_t* batch_row(const BatchSrc& b, int64_t B, int64_t a) {
  return b.batch ? b.batch + 3 * a : b.kg + 3 * b.index[b.starts[b.cursor[0]] * B + a];
}

// cand_rows / cand_cnt (optional): the batch BEFORE oversampling -- rows [0,B) the positives, rows [B, B+m) the generated
// negatives -- and how often random.choices picked each of them (cand_cnt zeroed by the caller of the kernel): the batch in
// folded form, a by-product of drawing it.
__global__ __launch_bounds__(256) void py_assemble_kernel(const MTDev* __restrict__ st, const uint32_t* __restrict__ U,
                                                           BatchSrc batch, int64_t B, int64_t ns,
                                                           const int32_t* __restrict__ fs, const int32_t* __restrict__ fo,
                                                           int32_t* __restrict__ pos_out, int32_t* __restrict__ neg_out,
                                                           int32_t* __restrict__ cand_rows, int32_t* __restrict__ cand_cnt) {
  const int64_t m = B * ns;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * m) return;
  const int p0 = st->pos;
  if (i < m) {  // choices over the positives (len B) come first in the stream
    const int64_t a = py_choice(st, U, p0, i, (double)B);
    if (pos_out) {
      const int32_t* row = batch_row(batch, B, a);
      pos_out[3 * i + 0] = row[0];
      pos_out[3 * i + 1] = row[1];
      pos_out[3 * i + 2] = row[2];
    }
    if (cand_rows) {
      atomicAdd(cand_cnt + (i & (KGE_CAND_POS_LANES - 1)) * B + a, 1);  // striped: B bins take m picks, ~ns hits per bin
      if (i < B) {
        const int32_t* r = batch_row(batch, B, i);
        cand_rows[3 * i + 0] = r[0]; cand_rows[3 * i + 1] = r[1]; cand_rows[3 * i + 2] = r[2];
      }
      int32_t* g = cand_rows + 3 * (B + i);  // generated negative i (generate_negative's row i, utils.py:114-121)
      g[0] = fs[i]; g[1] = batch_row(batch, B, i / ns)[1]; g[2] = fo[i];
    }
  } else {      // then over the negatives (len m); row j of generate_negative carries the relation of positive j // ns
    const int64_t j = py_choice(st, U, p0, i, (double)m);
    const int64_t o = i - m;
    if (neg_out) {
      neg_out[3 * o + 0] = fs[j];
      neg_out[3 * o + 1] = batch_row(batch, B, j / ns)[1];
      neg_out[3 * o + 2] = fo[j];
    }
    if (cand_rows) atomicAdd(cand_cnt + KGE_CAND_POS_LANES * B + j, 1);
  }
}

// New state after consuming `c` words: the block of U the consumption stopped in (NumPy/CPython keep pos = 624 at a
// block end and regenerate lazily, which is the representation kept here).
struct AdvanceArgs {
  MTDev* st;
  const uint32_t* U;
  int64_t L_u;
  const int64_t* consumed_ptr;  // data-dependent word count (NumPy stream), or null
  int64_t consumed_fixed;       // fixed word count (Python stream)
};

// One workgroup per generator.  The last kernel of a batch: also moves the batch cursor on (when there is one).
__global__ __launch_bounds__(256) void mt_advance_kernel(AdvanceArgs a0, AdvanceArgs a1, int64_t* __restrict__ cursor,
                                                          uint32_t* __restrict__ epoch, uint32_t* __restrict__ err) {
  __shared__ int64_t c_s;
  const AdvanceArgs& A = blockIdx.x == 0 ? a0 : a1;
  if (blockIdx.x == 1 && threadIdx.x == 0 && cursor) cursor[0] += 1;
  MTDev* st = A.st;
  if (threadIdx.x == 0 && st) {
    c_s = A.consumed_fixed;
    if (A.consumed_ptr) c_s = A.consumed_ptr[1] == (int64_t)(epoch[0] + 1u) ? A.consumed_ptr[0] : -1;  // -1: the 2m-th draw was not reached
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) epoch[0] += 1;  // this batch is done (block 0 owns the NumPy side, the only reader)
  if (!st) return;
  int64_t c = c_s;
  const int p0 = st->pos;
  if (A.consumed_ptr && c < 0) {  // fewer than the needed accepted draws in the buffer
    if (threadIdx.x == 0) atomicOr(err, 2u);
    c = (624 - p0) + A.L_u;
  }
  const int64_t a = p0 + c;  // absolute position counted from the start of the current state block
  if (a <= 624) {
    __syncthreads();
    if (threadIdx.x == 0) st->pos = (int32_t)a;
    return;
  }
  const int64_t idx = a - 624;             // words consumed out of U (idx >= 1)
  const int64_t blk = (idx - 1) / 624;     // block holding the last consumed word
  const int newpos = (int)(idx - blk * 624);
  const uint32_t* src = A.U + blk * 624;
  uint32_t tmp[3];
  for (int i = threadIdx.x, k = 0; i < 624; i += 256, ++k) tmp[k] = src[i];
  __syncthreads();
  for (int i = threadIdx.x, k = 0; i < 624; i += 256, ++k) st->mt[i] = tmp[k];
  if (threadIdx.x == 0) st->pos = newpos;
}


__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
struct SamplerPlan {
  int64_t K_np, K_py, ntiles;
  uint32_t* U_np;
  uint32_t* U_py;
  unsigned long long* tile_agg;
  uint32_t* epoch;
  int64_t* consumed;
  int32_t* fs;
  int32_t* fo;
  uint32_t* Jst;      // [2][kSeg][kJumpSplit][kJRow] partial results of the jumps (XOR over the slices = segment-start state)
  int segb_np, segb_py;  // blocks per segment (0 = the stream is generated sequentially)
  int64_t bytes;
};

static uint64_t smear64(uint64_t v) {
  v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32;
  return v;
}

static SamplerPlan plan_sampler(int64_t B, int64_t ns, int64_t num_e, void* ws) {
  SamplerPlan p;
  const int64_t m = B * ns;
  const uint64_t rng = (uint64_t)(num_e - 1), mask = smear64(rng);
  const double pacc = rng ? (double)(rng + 1) / (double)(mask + 1) : 1.0;
  const int64_t raw_np = rng ? (int64_t)(2.0 * m / pacc * 1.03) + 8192 : 0;
  p.K_np = rng ? cdiv(raw_np, 624) + 1 : 0;
  p.K_py = cdiv(4 * m, 624) + 1;
  p.ntiles = cdiv(624 + p.K_np * 624, kTileW);
  char* q = reinterpret_cast<char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += align_up(bytes, 256); return r; };
  p.U_np = (uint32_t*)take(std::max<int64_t>(1, p.K_np) * 624 * 4);
  p.U_py = (uint32_t*)take(p.K_py * 624 * 4);
  p.tile_agg = (unsigned long long*)take(p.ntiles * 8);
  p.epoch = (uint32_t*)take(256);
  p.consumed = (int64_t*)take(256);
  p.fs = (int32_t*)take(m * 4);
  p.fo = (int32_t*)take(m * 4);
  p.Jst = (uint32_t*)take((int64_t)2 * kSeg * kJumpSplit * kJRow * 4);  // partial jump results [2][kSeg][kJumpSplit][kJRow]
  auto segb = [](int64_t K) { const int64_t sb = cdiv(K, kSeg); return (int)(sb > kJumpBlocks ? sb : 0); };
  p.segb_np = segb(p.K_np);
  p.segb_py = segb(p.K_py);
  p.bytes = (q - reinterpret_cast<char*>(ws)) + 256;
  return p;
}

int64_t sampler_workspace_bytes(int64_t B, int64_t ns, int64_t num_e) { return plan_sampler(B, ns, num_e, nullptr).bytes + 512; }

// Fills polys_host[2][kSeg-1][312] (uint64) with the jump polynomials of the segment starts for this batch shape.
int sampler_plan_host(int64_t B, int64_t ns, int64_t num_e, uint64_t* polys_host) {
  const SamplerPlan p = plan_sampler(B, ns, num_e, nullptr);
  for (int z = 0; z < 2; ++z) {
    const int segb = z == 0 ? p.segb_np : p.segb_py;
    for (int s = 1; s < kSeg; ++s) {
      uint64_t* out = polys_host + ((int64_t)z * (kSeg - 1) + (s - 1)) * kPolyWords;
      if (segb == 0) { std::memset(out, 0, kPolyWords * 8); continue; }
      if (int rc = kge_mt_jump_poly((int64_t)624 * s * segb, out)) return rc;
    }
  }
  return KGE_OK;
}
int64_t sampler_plan_bytes() { return (int64_t)2 * (kSeg - 1) * kPolyWords * 8; }

// batch != null: the positives as a (B,3) array.  Otherwise rows index[starts[*cursor] * B + i] of kg, and *cursor += 1 at the end.
int build_batch_device(void* np_state, void* py_state, const int32_t* batch, const int32_t* kg, const int64_t* index, const int64_t* starts,
                       int64_t* cursor, int64_t B, int64_t ns, int64_t num_e, int32_t* pos_out, int32_t* neg_out, int32_t* cand_rows,
                       int32_t* cand_cnt, const void* plan_dev, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st) {
  if (wbytes < sampler_workspace_bytes(B, ns, num_e))
    return fail(KGE_EWORKSPACE, "kge_build_batch_device: workspace has %lld bytes, needs %lld", (long long)wbytes,
                (long long)sampler_workspace_bytes(B, ns, num_e));
  const SamplerPlan p = plan_sampler(B, ns, num_e, ws);
  const int64_t m = B * ns;
  MTDev* nps = static_cast<MTDev*>(np_state);
  MTDev* pys = static_cast<MTDev*>(py_state);
  const uint32_t rng = (uint32_t)(num_e - 1), mask = (uint32_t)smear64(rng);
  ProfScope prof(KGE_PROF_SAMPLER, st);
  const int64_t n_cnt = cand_cnt ? KGE_CAND_POS_LANES * B + m : 0;
  if (cand_cnt && !rng) hipLaunchKernelGGL(fill_i32_kernel, dim3((int)cdiv(n_cnt, 256)), dim3(256), 0, st, cand_cnt, n_cnt, 0);
  const int sb_np = plan_dev ? p.segb_np : 0, sb_py = plan_dev ? p.segb_py : 0;
  hipLaunchKernelGGL(mt_generate_kernel, dim3(2), dim3(256), 0, st, nps, pys, p.U_np, sb_np ? kJumpBlocks : (int)p.K_np, p.U_py,
                     sb_py ? kJumpBlocks : (int)p.K_py, (sb_np || sb_py) ? p.Jst : nullptr);
  if (sb_np || sb_py) {
    hipLaunchKernelGGL(mt_jump_kernel, dim3(10 * kJumpSplit, kSeg - 1, 2), dim3(256), 0, st, nps, pys, p.U_np, p.U_py, static_cast<const uint64_t*>(plan_dev),
                       p.Jst, sb_np, sb_py);
    hipLaunchKernelGGL(mt_generate_seg_kernel, dim3(kSeg, 2), dim3(256), 0, st, p.U_np, (int)p.K_np, sb_np, p.U_py, (int)p.K_py, sb_py, p.Jst);
  }
  if (rng) {
    const int64_t L_u = p.K_np * 624;
    hipLaunchKernelGGL(np_select_kernel, dim3((int)p.ntiles), dim3(256), 0, st, nps, p.U_np, L_u, rng, mask, p.tile_agg, p.epoch, m, p.fs, p.fo,
                       p.consumed, cand_cnt, n_cnt);
  } else {  // a range of one: numpy draws nothing
    hipLaunchKernelGGL(fill_i32_kernel, dim3((int)cdiv(m, 256)), dim3(256), 0, st, p.fs, m, 0);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((int)cdiv(m, 256)), dim3(256), 0, st, p.fo, m, 0);
  }
  hipLaunchKernelGGL(py_assemble_kernel, dim3((int)cdiv(2 * m, 256)), dim3(256), 0, st, pys, p.U_py, BatchSrc{batch, kg, index, starts, cursor}, B, ns, p.fs, p.fo, pos_out,
                     neg_out, cand_rows, cand_cnt);
  const AdvanceArgs adv_np{rng ? nps : nullptr, p.U_np, p.K_np * 624, p.consumed, 0};
  const AdvanceArgs adv_py{pys, p.U_py, p.K_py * 624, nullptr, 4 * m};
  hipLaunchKernelGGL(mt_advance_kernel, dim3(2), dim3(256), 0, st, adv_np, adv_py, batch ? nullptr : cursor, p.epoch, err);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}


// =================================================================================================================
// Streamed form of the same sampler (kge_sampler_stream_*): word generation runs AHEAD of the batches that consume it.
//
// Drawing a batch the way above puts ~70 us of latency-bound generator kernels (sequential pre-pass, GF(2) jumps, segment
// regeneration) in front of every select / assemble pair.  But an MT19937 stream is one fixed sequence of 624-word blocks:
// which words a batch consumes is data dependent (NumPy's masked rejection), the words themselves are not.  Each generator
// therefore gets a RING of raw state blocks in HBM that a producer extends round by round -- 16 segments of `segb` blocks per
// round, started from block head-1 by the same jump polynomials -- on its own stream, one batch ahead, while the consumer side
// of a batch shrinks to select -> assemble -> advance.  Global numbering: block 0 is the state block the ring was seeded with,
// word w of the stream lives in block w / 624 at offset w % 624, block b is stored in slot b % RB.
//   control word per generator: head (blocks [0, head) exist; committed lazily by the next round), wpos (next word to consume).
//   A round covers R = kStreamRound batches (one set of jumps per R batches).  Rounds are offered one at the start and then one
//   in front of every batch k = 0, R, 2R, ...; the draw of batch k waits only for the rounds offered BEFORE the one in front of
//   it, so a round has R batches' time to finish -- it never sits on a batch's critical path.  A round generates only while fewer
//   than LOW = (2R+1) M words lie between wpos and the head (M = the most a batch can consume; P*624 >= R M words per round):
//     invariant   when batch k = 0 mod R starts, the completed rounds hold >= R M words ahead of wpos (batches k .. k+R-1)
//     skip        avail_seen >= (2R+1) M; until batch k+R starts at most R+1 batches are consumed (R + one the producer may not
//                 have seen yet)                                                                   -> >= R M words then
//     generate    >= R M (invariant) - R M consumed + P*624 >= R M new                              -> >= R M words then
//   (the caller orders the round in front of batch k behind the consumer of batch k-2, so wpos is stale by at most one batch).
//   Ring capacity RB >= (LOW + 2 P*624) / 624 + 8 blocks, rounded up to a power of two: nothing still readable is overwritten,
//   the block holding the last consumed word (the generator state handed back to NumPy / CPython) included.
struct GenCtl {
  int64_t head;        // blocks generated and committed
  int64_t wpos;        // global index of the next word to consume
  int64_t head_round;  // head when the running / last round started
  int32_t go;          // the last round generated (its blocks are committed by the next round's first kernel)
  int32_t pad;
  int64_t wsnap;       // wpos as the running batch found it (the advance of a batch shares a launch with its assembly)
  int64_t csnap;       // the batch cursor likewise (kept in the Python generator's control block)
};
struct Ring {
  uint32_t* blocks;
  int64_t RB;      // capacity in blocks (a power of two)
  GenCtl* ctl;
  int64_t P;       // blocks per round
  int64_t low;     // words: generate while avail < low
  int32_t segb;    // blocks per segment (0: the round is one sequential pass of P blocks)
};
__device__ __forceinline__ uint32_t* ring_block(const Ring& r, int64_t gb) { return r.blocks + (gb & (r.RB - 1)) * 624; }  // RB is a power of two
__device__ __forceinline__ uint32_t ring_word(const Ring& r, int64_t w) {
  const int64_t b = w / 624;
  return mt_temper(r.blocks[(b & (r.RB - 1)) * 624 + (w - b * 624)]);
}

// Seeds both rings from the generator state blobs: block 0 = the state block, consumption starts at its position.
__global__ __launch_bounds__(256) void ring_init_kernel(Ring r0, const MTDev* __restrict__ st0, Ring r1, const MTDev* __restrict__ st1) {
  const Ring& r = blockIdx.x == 0 ? r0 : r1;
  const MTDev* st = blockIdx.x == 0 ? st0 : st1;
  for (int i = threadIdx.x; i < 624; i += 256) r.blocks[i] = st->mt[i];
  if (threadIdx.x == 0) { r.ctl->head = 1; r.ctl->wpos = st->pos; r.ctl->head_round = 1; r.ctl->go = 0; r.ctl->wsnap = st->pos; r.ctl->csnap = 0; }
}

// Same as mt_regen_blocks, into ring slots: regenerates global blocks [g_begin, g_end) from the block held in buf[0].
#define KGE_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ void ring_regen_blocks(uint32_t (*buf)[624], const Ring& r, int64_t g_begin, int64_t g_end) {
  const int tid = threadIdx.x;
  int cur = 0;
  for (int64_t g = g_begin; g < g_end; ++g) {
    const uint32_t* o = buf[cur];
    uint32_t* n = buf[cur ^ 1];
    uint32_t* dst = ring_block(r, g);
    if (tid < 227) {
      const uint32_t a0 = o[tid], a1 = o[tid + 1], a397 = o[tid + 397];
      const uint32_t b0 = o[227 + tid], b1 = o[228 + tid];
      const uint32_t v0 = a397 ^ mt_twist(a0, a1);
      const uint32_t v1 = v0 ^ mt_twist(b0, b1);
      n[tid] = v0;
      n[227 + tid] = v1;
      dst[tid] = v0;
      dst[227 + tid] = v1;
      if (tid < 169) {
        const uint32_t v2 = v1 ^ mt_twist(o[454 + tid], o[455 + tid]);
        n[454 + tid] = v2;
        dst[454 + tid] = v2;
      } else if (tid == 169) {
        const uint32_t n0 = o[397] ^ mt_twist(o[0], o[1]);
        const uint32_t v2 = v1 ^ mt_twist(o[623], n0);
        n[623] = v2;
        dst[623] = v2;
      }
    }
    KGE_LDS_BARRIER();
    cur ^= 1;
  }
}
#undef KGE_LDS_BARRIER

// First kernel of a round (one workgroup per generator): commits the previous round, decides go / skip, and -- when going --
// regenerates the first kJumpBlocks blocks behind the head sequentially (all P of them when the round is not segmented).
__global__ __launch_bounds__(256) void ring_pre_kernel(Ring r0, Ring r1, uint32_t* __restrict__ Jst) {
  __shared__ uint32_t buf[2][624];
  __shared__ int64_t head_s;
  __shared__ int go_s;
  const Ring& r = blockIdx.x == 0 ? r0 : r1;
  __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x == 0) {
    GenCtl* c = r.ctl;
    int64_t head = c->head;
    if (c->go) head = c->head_round + r.P;  // the previous round's blocks are complete (same stream): commit them
    const int64_t wpos = __hip_atomic_load(&c->wpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a consumer may be advancing it)
    const int go = head * 624 - wpos < r.low ? 1 : 0;
    c->head = head;
    c->head_round = head;
    c->go = go;
    head_s = head;
    go_s = go;
  }
  (void)Jst;
  __syncthreads();
  if (!go_s) return;
  const int64_t head = head_s;
  const uint32_t* src = ring_block(r, head - 1);
  for (int i = threadIdx.x; i < 624; i += 256) buf[0][i] = src[i];
  __syncthreads();
  ring_regen_blocks(buf, r, head, head + (r.segb ? (int64_t)kJumpBlocks : r.P));
}

// The jumps of a round: W = block head-1 followed by the blocks behind it (jump_core).  grid (10 * kJumpSplit, kSeg - 1, 2).
__global__ __launch_bounds__(256) void ring_jump_kernel(Ring r0, Ring r1, const uint64_t* __restrict__ polys, uint32_t* __restrict__ Jpart) {
  const int z = blockIdx.z;
  const Ring& r = z == 0 ? r0 : r1;
  if (r.segb == 0 || !r.ctl->go) return;
  const int64_t base = r.ctl->head_round - 1;  // global block of W's first word
  jump_core([&](int idx) { const int b = idx / 624; return ring_block(r, base + b)[idx - b * 624]; }, polys + (int64_t)z * (kSeg - 1) * kPolyWords,
            Jpart + (int64_t)z * kSeg * kJumpSplit * kJRow);
}

// grid (kSeg, 2): segment s of generator z regenerates blocks [head + s*segb, head + (s+1)*segb) of the round.
__global__ __launch_bounds__(256) void ring_seg_kernel(Ring r0, Ring r1, const uint32_t* __restrict__ Jst) {
  __shared__ uint32_t buf[2][624];
  const int z = blockIdx.y, s = blockIdx.x;
  const Ring& r = z == 0 ? r0 : r1;
  if (r.segb == 0 || !r.ctl->go) return;
  const int64_t head = r.ctl->head_round;
  int64_t b0 = (int64_t)s * r.segb, b1 = min(r.P, (int64_t)(s + 1) * r.segb);
  if (s == 0) b0 = kJumpBlocks;
  if (b0 >= b1) return;
  __builtin_amdgcn_s_setprio(3);
  const uint32_t* last = ring_block(r, head + kJumpBlocks - 1);
  for (int i = threadIdx.x; i < 624; i += 256) buf[0][i] = s == 0 ? last[i] : jump_state_word(Jst, z, s, i);
  __syncthreads();
  ring_regen_blocks(buf, r, head + b0, head + b1);
}

// np_select_kernel on ring words: tiles of 1024 words from wpos on, at most max_words of them.
__device__ __forceinline__ void ring_snapshot(GenCtl* c_np, GenCtl* c_py, const int64_t* cursor) {
  if (c_np) c_np->wsnap = c_np->wpos;
  c_py->wsnap = c_py->wpos;
  c_py->csnap = cursor ? cursor[0] : 0;
}
__global__ void ring_snapshot_kernel(GenCtl* c_py, const int64_t* cursor) { ring_snapshot(nullptr, c_py, cursor); }

__global__ __launch_bounds__(256) void ring_select_kernel(Ring r, int64_t max_words, uint32_t rng, uint32_t mask, unsigned long long* tile_agg,
                                                           const uint32_t* __restrict__ epoch_ptr, int64_t m, int32_t* __restrict__ fs,
                                                           int32_t* __restrict__ fo, int64_t* __restrict__ consumed,
                                                           int32_t* __restrict__ zero_p, int64_t zero_n, GenCtl* c_py,
                                                           const int64_t* __restrict__ cursor) {
  __shared__ int wsum[4];
  __shared__ long long pre_s[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) ring_snapshot(r.ctl, c_py, cursor);  // what the assembly launch reads while its last blocks advance
  for (int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; z < zero_n; z += (int64_t)gridDim.x * blockDim.x) zero_p[z] = 0;
  const unsigned long long ep = (unsigned long long)(epoch_ptr[0] + 1u);
  const int64_t w0 = r.ctl->wpos;
  const int64_t base = (int64_t)blockIdx.x * kTileW + threadIdx.x * 4;
  uint32_t v[4];
  int a[4], c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = base + j < max_words ? (ring_word(r, w0 + base + j) & mask) : 0xFFFFFFFFu;
    a[j] = (base + j < max_words && v[j] <= rng) ? 1 : 0;
    c += a[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = c;
  for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(inc, d, 64); if (lane >= d) inc += up; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(tile_agg + blockIdx.x, ep << 32 | (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  long long before = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) {
    unsigned long long s;
    do { s = __hip_atomic_load(tile_agg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((s >> 32) != ep);
    before += (long long)(s & 0xFFFFFFFFull);
  }
  for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) pre_s[wave] = before;
  __syncthreads();
  int64_t rk = pre_s[0] + pre_s[1] + pre_s[2] + pre_s[3] + inc - c;
  for (int w = 0; w < wave; ++w) rk += wsum[w];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!a[j]) continue;
    if (rk < m) fs[rk] = (int32_t)v[j];
    else if (rk < 2 * m) fo[rk - m] = (int32_t)v[j];
    if (rk == 2 * m - 1) { consumed[0] = base + j + 1; consumed[1] = (int64_t)ep; }
    ++rk;
  }
}

__device__ __forceinline__ int64_t ring_choice(const Ring& r, int64_t w0, int64_t i, double n) {
  const uint32_t a = ring_word(r, w0 + 2 * i) >> 5, b = ring_word(r, w0 + 2 * i + 1) >> 6;
  const double x = ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
  return (int64_t)floor(x * n);
}

// New generator state after a batch (one workgroup): wpos += words consumed; the state blob handed back to NumPy / CPython = the
// block holding the last consumed word + the position behind it.
__device__ __forceinline__ void ring_advance(const Ring& r, MTDev* st, int64_t c) {
  const int64_t w1 = r.ctl->wsnap + c;  // next word to consume
  if (c > 0) {
    const int64_t blk = (w1 - 1) / 624;  // block holding the last consumed word
    const uint32_t* src = ring_block(r, blk);
    for (int i = threadIdx.x; i < 624; i += 256) st->mt[i] = src[i];
    if (threadIdx.x == 0) st->pos = (int32_t)(w1 - blk * 624);
  }
  if (threadIdx.x == 0) __hip_atomic_store(&r.ctl->wpos, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// py_assemble_kernel on ring words; the two workgroups behind the assembly grid advance the generators (they depend on the select
// kernel only, like the assembly itself): the batch cursor and the batch epoch move on there.  Everything the assembly reads that
// the advance writes (wpos, cursor) is read from the snapshot the select kernel took.
struct AdvArgs {
  Ring r_np; MTDev* st_np; int64_t max_np; const int64_t* consumed;  // st_np null: a range of one, NumPy draws nothing
  MTDev* st_py; int64_t consumed_py; int64_t* cursor; uint32_t* epoch; uint32_t* err;
};
__global__ __launch_bounds__(256) void ring_assemble_kernel(Ring r, BatchSrc batch, int64_t B, int64_t ns, const int32_t* __restrict__ fs,
                                                             const int32_t* __restrict__ fo, int32_t* __restrict__ pos_out,
                                                             int32_t* __restrict__ neg_out, int32_t* __restrict__ cand_rows,
                                                             int32_t* __restrict__ cand_cnt, int nblk, AdvArgs adv) {
  __shared__ int64_t c_s;
  if ((int)blockIdx.x >= nblk) {
    if ((int)blockIdx.x == nblk) {  // NumPy generator
      if (threadIdx.x == 0) {
        c_s = adv.st_np ? (adv.consumed[1] == (int64_t)(adv.epoch[0] + 1u) ? adv.consumed[0] : -1) : 0;  // -1: the 2m-th accepted draw was not reached
        if (adv.st_np && c_s < 0) { atomicOr(adv.err, 2u); c_s = adv.max_np; }
      }
      __syncthreads();
      if (threadIdx.x == 0) adv.epoch[0] += 1;  // this batch is done (the select kernel is the only reader)
      if (adv.st_np) ring_advance(adv.r_np, adv.st_np, c_s);
    } else {                        // Python generator + batch cursor
      if (threadIdx.x == 0 && adv.cursor) adv.cursor[0] += 1;
      ring_advance(r, adv.st_py, adv.consumed_py);
    }
    return;
  }
  const int64_t m = B * ns;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * m) return;
  const int64_t w0 = r.ctl->wsnap;
  if (i < m) {
    const int64_t a = ring_choice(r, w0, i, (double)B);
    if (pos_out) {
      const int32_t* row = batch_row(batch, B, a);
      pos_out[3 * i + 0] = row[0];
      pos_out[3 * i + 1] = row[1];
      pos_out[3 * i + 2] = row[2];
    }
    if (cand_rows) {
      atomicAdd(cand_cnt + (i & (KGE_CAND_POS_LANES - 1)) * B + a, 1);
      if (i < B) {
        const int32_t* q = batch_row(batch, B, i);
        cand_rows[3 * i + 0] = q[0]; cand_rows[3 * i + 1] = q[1]; cand_rows[3 * i + 2] = q[2];
      }
      int32_t* g = cand_rows + 3 * (B + i);
      g[0] = fs[i]; g[1] = batch_row(batch, B, i / ns)[1]; g[2] = fo[i];
    }
  } else {
    const int64_t j = ring_choice(r, w0, i, (double)m);
    const int64_t o = i - m;
    if (neg_out) {
      neg_out[3 * o + 0] = fs[j];
      neg_out[3 * o + 1] = batch_row(batch, B, j / ns)[1];
      neg_out[3 * o + 2] = fo[j];
    }
    if (cand_rows) atomicAdd(cand_cnt + KGE_CAND_POS_LANES * B + j, 1);
  }
}

struct StreamPlan {
  SamplerPlan sp;   // tile_agg / epoch / consumed / fs / fo / Jst of the workspace, K_* and segb_* of the batch shape
  Ring r_np, r_py;
  int64_t max_np;   // most words of the NumPy stream one batch may consume (the old raw buffer's size)
  int64_t bytes;
};

#ifndef KGE_STREAM_ROUND
#define KGE_STREAM_ROUND 4
#endif
constexpr int kStreamRound = KGE_STREAM_ROUND;  // batches per production round

static StreamPlan plan_stream(int64_t B, int64_t ns, int64_t num_e, void* buf) {
  StreamPlan q;
  q.sp = plan_sampler(B, ns, num_e, buf);
  const int64_t m = B * ns;
  char* p = reinterpret_cast<char*>(((uintptr_t)(buf ? reinterpret_cast<char*>(buf) : (char*)nullptr) + (uintptr_t)align_up(q.sp.bytes, 256) + 255) & ~(uintptr_t)255);
  auto ring = [&](int64_t K, int64_t max_words) {
    Ring r;
    const int64_t Kr = std::max<int64_t>(K, 1) * kStreamRound;  // blocks one round must cover
    const int64_t sb = cdiv(Kr, kSeg);
    r.segb = sb > kJumpBlocks ? (int32_t)sb : 0;
    r.P = r.segb ? (int64_t)kSeg * r.segb : Kr;
    const int64_t M = std::max<int64_t>(max_words, 1);
    r.low = (2 * kStreamRound + 1) * M;
    int64_t rb = (r.low + 2 * r.P * 624) / 624 + 8, pow2 = 64;
    while (pow2 < rb) pow2 <<= 1;
    r.RB = pow2;
    r.ctl = reinterpret_cast<GenCtl*>(p);
    p += 256;
    r.blocks = reinterpret_cast<uint32_t*>(p);
    p += align_up(r.RB * 624 * 4, 256);
    return r;
  };
  q.max_np = q.sp.K_np * 624;
  q.r_np = ring(q.sp.K_np, q.max_np);
  q.r_py = ring(q.sp.K_py, 4 * m);
  q.bytes = (p - (buf ? reinterpret_cast<char*>(buf) : (char*)nullptr)) + 512;
  return q;
}

int64_t sampler_stream_bytes(int64_t B, int64_t ns, int64_t num_e) { return plan_stream(B, ns, num_e, nullptr).bytes + 512; }
int sampler_stream_round() { return kStreamRound; }

// Jump polynomials of the streamed form's rounds: polys_host[2][kSeg-1][312] (uint64), all zero for a generator whose rounds are
// one sequential pass.
int sampler_stream_plan_host(int64_t B, int64_t ns, int64_t num_e, uint64_t* polys_host) {
  const StreamPlan q = plan_stream(B, ns, num_e, nullptr);
  for (int z = 0; z < 2; ++z) {
    const int segb = z == 0 ? q.r_np.segb : q.r_py.segb;
    for (int s = 1; s < kSeg; ++s) {
      uint64_t* out = polys_host + ((int64_t)z * (kSeg - 1) + (s - 1)) * kPolyWords;
      if (segb == 0) { std::memset(out, 0, kPolyWords * 8); continue; }
      if (int rc = kge_mt_jump_poly((int64_t)624 * s * segb, out)) return rc;
    }
  }
  return KGE_OK;
}

int sampler_stream_init(void* buf, int64_t bytes, const void* np_state, const void* py_state, int64_t B, int64_t ns, int64_t num_e, hipStream_t st) {
  if (bytes < sampler_stream_bytes(B, ns, num_e))
    return fail(KGE_EWORKSPACE, "kge_sampler_stream_init: buffer has %lld bytes, needs %lld", (long long)bytes, (long long)sampler_stream_bytes(B, ns, num_e));
  const StreamPlan q = plan_stream(B, ns, num_e, buf);
  hipLaunchKernelGGL(ring_init_kernel, dim3(2), dim3(256), 0, st, q.r_np, static_cast<const MTDev*>(np_state), q.r_py, static_cast<const MTDev*>(py_state));
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// One production round for both generators (skipped on the device while enough words lie ahead of the consumers).
int sampler_stream_produce(void* buf, int64_t bytes, int64_t B, int64_t ns, int64_t num_e, const void* plan_dev, hipStream_t st) {
  if (bytes < sampler_stream_bytes(B, ns, num_e)) return fail(KGE_EWORKSPACE, "kge_sampler_stream_produce: buffer too small");
  StreamPlan q = plan_stream(B, ns, num_e, buf);
  if (!plan_dev) {  // no jump polynomials: every round is one sequential pass
    q.r_np.segb = 0; q.r_py.segb = 0;
  }
  const bool rng = num_e > 1;
  if (!rng) { q.r_np.P = 0; q.r_np.low = 0; }  // a range of one: NumPy draws nothing
  ProfScope prof(KGE_PROF_SAMPLER, st);
  hipLaunchKernelGGL(ring_pre_kernel, dim3(2), dim3(256), 0, st, q.r_np, q.r_py, q.sp.Jst);
  if (q.r_np.segb || q.r_py.segb) {
    hipLaunchKernelGGL(ring_jump_kernel, dim3(10 * kJumpSplit, kSeg - 1, 2), dim3(256), 0, st, q.r_np, q.r_py, static_cast<const uint64_t*>(plan_dev), q.sp.Jst);
    hipLaunchKernelGGL(ring_seg_kernel, dim3(kSeg, 2), dim3(256), 0, st, q.r_np, q.r_py, q.sp.Jst);
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// Select + assemble + advance of one batch from the rings (the consumer half of kge_build_batch_device_indexed).
int sampler_stream_draw(void* buf, int64_t bytes, void* np_state, void* py_state, const int32_t* batch, const int32_t* kg, const int64_t* index,
                        const int64_t* starts, int64_t* cursor, int64_t B, int64_t ns, int64_t num_e, int32_t* pos_out, int32_t* neg_out,
                        int32_t* cand_rows, int32_t* cand_cnt, uint32_t* err, hipStream_t st) {
  if (bytes < sampler_stream_bytes(B, ns, num_e)) return fail(KGE_EWORKSPACE, "kge_sampler_stream_draw: buffer too small");
  const StreamPlan q = plan_stream(B, ns, num_e, buf);
  const SamplerPlan& p = q.sp;
  const int64_t m = B * ns;
  const uint32_t rng = (uint32_t)(num_e - 1), mask = (uint32_t)smear64(rng);
  ProfScope prof(KGE_PROF_SAMPLER, st);
  const int64_t n_cnt = cand_cnt ? KGE_CAND_POS_LANES * B + m : 0;
  const int64_t* cur = batch ? nullptr : cursor;
  if (rng) {
    const int ntiles = (int)cdiv(q.max_np, kTileW);
    hipLaunchKernelGGL(ring_select_kernel, dim3(ntiles), dim3(256), 0, st, q.r_np, q.max_np, rng, mask, p.tile_agg, p.epoch, m, p.fs, p.fo, p.consumed,
                       cand_cnt, n_cnt, q.r_py.ctl, cur);
  } else {
    hipLaunchKernelGGL(ring_snapshot_kernel, dim3(1), dim3(1), 0, st, q.r_py.ctl, cur);
    if (cand_cnt) hipLaunchKernelGGL(fill_i32_kernel, dim3((int)cdiv(n_cnt, 256)), dim3(256), 0, st, cand_cnt, n_cnt, 0);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((int)cdiv(m, 256)), dim3(256), 0, st, p.fs, m, 0);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((int)cdiv(m, 256)), dim3(256), 0, st, p.fo, m, 0);
  }
  const int nblk = (int)cdiv(2 * m, 256);
  const AdvArgs adv{q.r_np, rng ? static_cast<MTDev*>(np_state) : nullptr, q.max_np, p.consumed, static_cast<MTDev*>(py_state), 4 * m,
                    batch ? nullptr : cursor, p.epoch, err};
  // (the assembly reads the cursor from the snapshot: its last workgroup moves the real one on)
  hipLaunchKernelGGL(ring_assemble_kernel, dim3(nblk + 2), dim3(256), 0, st, q.r_py, BatchSrc{batch, kg, index, starts, batch ? nullptr : &q.r_py.ctl->csnap}, B, ns,
                     p.fs, p.fo, pos_out, neg_out, cand_rows, cand_cnt, nblk, adv);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
