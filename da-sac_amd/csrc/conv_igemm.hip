// Implicit-GEMM convolution on the gfx950 matrix cores: exact fp32 (v_mfma_f32_32x32x2_f32, the default) or,
// opt-in per call (template flag X3), split-bf16 products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Replaces every nn.Conv2d call of the reference backbone (models/deeplabv2.py:59,65-66,70,107,
// 122,147-148,262-263; models/fcn.py:49,53,57,78,88) -- 1x1, dilated 3x3, 7x7/2 stem, and the
// four-branch ASPP sum as ONE contraction -- for forward, data-gradient and weight-gradient.
//
//   forward / dgrad :  Out[m][pix] = sum_k  Wp[k][m] * gather(X, k, pix)        (conv_gemm)
//   wgrad           :  G[m][k]     = sum_pix dZ[m][pix] * gather(X, k, pix)     (conv_wgrad)
//
// `gather` is table driven: row k of the im2col matrix is (channel plane offset, dh, dw), so one
// kernel covers any kernel size / dilation / padding / multi-branch layout; the table row is
// wave-uniform (scalar loads), lanes run along pixels (coalesced NCHW reads).  Tiles go through
// LDS (k-interleaved 16-byte words, one conflict-free ds_read_b128 per operand and 4 MFMAs), register-staged
// double buffering: the global loads of K-step t+1 are in flight while the MFMAs of step t issue.  The BN(eval) scale/shift, bias, residual add, ReLU and the ReLU-mask of the backward
// pass are folded into the epilogue ("ABN": conv+BN+ReLU in one pass, no extra HBM round trip).
//
// Roofline: MFMA-bound, 2*M*Npix*K flops per launch against the 157.3 TFLOP/s fp32 matrix peak
// (X3: three bf16 MFMAs per product -> 2500/3 TFLOP/s equivalent; today limited by the fp32 -> LDS staging).
#include "common.hpp"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace dasac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 16;            // K-step of the forward/dgrad kernel
constexpr int kThreads = 256;      // 4 waves
constexpr int kInvalid = 1 << 20;  // dh of a padded table row: never in bounds
constexpr int kSkWorkersPerCu = 3;                       // persistent 128x128 workers per CU (<=168 registers, 32 KB LDS each; four spill: EXPERIMENTS.md)
constexpr int kSkWorkers = kNumCu * kSkWorkersPerCu;     // 768, multiple of 8: the grid of every stream-K launch
constexpr int kTileSlots = kNumCu * 4;                   // resident blocks of the tile-per-block kernel (4 per CU)
constexpr int kTailSlots = 2048;                         // deposit / flag slots of the split-K tail (SK == 2): 128 MB of workspace

#ifdef DASAC_TRACE_TILES
// Diagnostic build only (tools/tile_timeline.py): per tile-per-block workgroup four s_memtime stamps -- start, first tile in LDS,
// end of the K loop, end of the epilogue -- plus the hardware XCC / CU ids.  Never part of the shipped library.
__device__ unsigned long long* g_tile_trace = nullptr;
__device__ __forceinline__ void trace_stamp(int slot) {
  if (g_tile_trace && threadIdx.x == 0) g_tile_trace[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
}
#define DASAC_STAMP(slot) trace_stamp(slot)
#else
#define DASAC_STAMP(slot)
#endif

struct GemmGeom {
  // gathered tensor X [Nb, Cx, H, W]
  int H, W, CxHW;                // CxHW = Cx*H*W (image stride)
  // pixel grid of the GEMM's N dimension: Nb x OH x OW, gather coordinate = o*stride + d
  int OH, OW, stride;
  int Npix;                      // Nb*OH*OW
  int n_tile0;                   // first pixel tile of this launch (a conv may be issued as several launches over pixel ranges)
  int M, Mpad, Kpad;
  // output tensor [Nb, M, OutH, OutW]; element (oh*ostride, ow*ostride)
  int OutH, OutW, ostride;
  unsigned x_bytes, w_bytes, z_bytes, out_bytes;  // buffer-descriptor extents (gathered tensor, packed weights, dZ, output): < 4 GiB
};

struct Epilogue {
  const float* shift;   // [M] or null (the BN scale is folded into the packed weights)
  const float* res;     // same layout as out, or null  (added before ReLU)
  const float* mask;    // same layout as out, or null  (out = mask>0 ? out : 0, ReLU backward)
  int relu;
  // ReLU pattern as ONE BIT per element instead of a whole fp32 activation read back only to test `> 0`:
  //   bits[m][pix >> 5] bit (pix & 31), pix = flattened (n, oh, ow) index of the GEMM's pixel axis, w32 = ceil(Npix / 32) words
  //   per row.  A 32-pixel group of a tile is 32-aligned (tiles start at multiples of 128), so every word belongs to exactly one
  //   wave: the producer (forward conv with ReLU) writes obits from wave ballots, the consumer (the data-gradient GEMM whose
  //   output has the producer's shape) reads mbits -- 1/32 of the fp32 mask's bytes.
  const unsigned* mbits;   // consumer: out = bit ? out : 0    (exclusive with `mask`)
  unsigned* obits;         // producer: bit = out > 0 (after ReLU); ostride must be 1
  int w32;
  // BITS == 3 (batch-statistics BatchNorm behind this conv, deeplabv2.py:15 in baseline / AdaBN mode): the epilogue also
  // leaves, per 128-pixel tile, the per-channel sum and sum of squares of what it stores -- stats[pixel tile][0|1][Mpad] --
  // so that the stand-alone statistics pass over z (one full read of the activation per BN layer) disappears.  Every slot
  // of a tile is written by exactly one workgroup; a fixed-order sum over the tiles follows (bn_train_finalize): deterministic.
  float* stats;
};

// ------------------------------------------------------------------------------------------
// forward / dgrad kernel
// ------------------------------------------------------------------------------------------
// Buffer addressing.  Every global read of the GEMM loops goes through a raw buffer descriptor:
//   address = base + voffset (VGPR) + soffset (SGPR);  reads past num_records return 0.
// That turns the zero padding of the convolution into an ADDRESS decision (out-of-image taps get the
// poison offset 2^31 >= num_records) and moves the per-row part of the address (channel plane, K-step
// of the weights) into the scalar operand -- the vector ALU, which shares issue bandwidth with the
// matrix pipe (measured: ~6 matrix-pipe cycles lost per VALU instruction), stays almost idle.
// Offsets are UNSIGNED 32-bit byte offsets: a tensor may be up to 4 GiB - 4 KiB (round 5; rounds 1-4 kept them below 2^31 and
// used 2^31 as the poison).  The poison is the largest offset: always >= num_records, whatever scalar offset is added (the
// hardware compares the per-lane offset with num_records - soffset), and never combined with an instruction immediate.
constexpr unsigned kPoison = 0xFFFFFFFFu;
constexpr int64_t kMaxTensorBytes = (1ll << 32) - 4096;
constexpr int kRsrcFlags = 0x00020000;   // raw buffer, 32-bit data format (gfx9 family word 3)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, kRsrcFlags);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// agent-scope (sc1) 16-byte accesses for data handed from one workgroup to another inside a launch: an sc1 store is
// written through to memory, an sc1 load does not hit a stale line of this XCD's L2 / this CU's L1 -- the per-XCD L2s
// are not coherent with each other (MI355X_MICROARCH.md, inter-workgroup visibility).  With both sides sc1 the hand-off
// needs no buffer_wbl2 / buffer_inv fences (1.7-6.5 us each), only the flag's own release/acquire ordering.
constexpr int kAuxSc1 = 16;
__device__ __forceinline__ f32x4 buf_f32x4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kAuxSc1);
  return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}

// v_writelane_b32 (this clang exposes no __builtin_amdgcn_writelane; the LLVM intrinsic is reachable through its asm label, the way
// the HIP headers reach llvm.amdgcn.s.barrier).  Rounds 3-4 emitted the instruction from inline asm, which hides it from the
// hazard recogniser: gfx940+ needs two wait states between a VALU that writes an SGPR (the ballot's v_cmp) and a VALU that reads it,
// and only instruction-scheduling luck had put them there.  put_mask_rows: the two 32-bit halves of a wave ballot over accumulator
// register `rg` are the mask words of rows (rg&3) + 8*(rg>>2) (lanes 0-31) and 4 below (lanes 32-63) of a 32x32 MFMA tile: lane r
// of `acc` collects row r.
// Toolchain note: a compiler that has the builtin uses it; otherwise the intrinsic's overloaded name
// `llvm.amdgcn.writelane.i32` is the one of LLVM 19+ (ROCm >= 6.3; this image: ROCm 7.2) -- older toolchains spell it
// `llvm.amdgcn.writelane` and are not supported by this build ("toolchain").
#if __has_builtin(__builtin_amdgcn_writelane)
template <int LANE>
__device__ __forceinline__ int writelane_c(int val, int old) {
  return __builtin_amdgcn_writelane(val, LANE, old);
}
#else
extern "C" __device__ int dasac_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
template <int LANE>
__device__ __forceinline__ int writelane_c(int val, int old) {
  return dasac_llvm_writelane(val, LANE, old);
}
#endif
__device__ __forceinline__ int put_mask_rows(int rg, unsigned long long ballot, int acc) {
  const int lo = (int)(unsigned)ballot, hi = (int)(unsigned)(ballot >> 32);
  switch (rg) {      // rg is the index of a fully unrolled loop: the switch folds to one case
#define DASAC_ROW_(RG)                                               \
  case RG:                                                           \
    acc = writelane_c<(RG & 3) + 8 * (RG >> 2)>(lo, acc);            \
    acc = writelane_c<(RG & 3) + 8 * (RG >> 2) + 4>(hi, acc);        \
    break;
    DASAC_ROW_(0) DASAC_ROW_(1) DASAC_ROW_(2) DASAC_ROW_(3) DASAC_ROW_(4) DASAC_ROW_(5) DASAC_ROW_(6) DASAC_ROW_(7)
    DASAC_ROW_(8) DASAC_ROW_(9) DASAC_ROW_(10) DASAC_ROW_(11) DASAC_ROW_(12) DASAC_ROW_(13) DASAC_ROW_(14) DASAC_ROW_(15)
#undef DASAC_ROW_
  }
  return acc;
}

// Split-bf16 ("bf16x3") operands: x = head + tail with head = bf16(x) (round to nearest even) and
// tail = bf16(x - head); eight consecutive k values of one row / pixel make one 16-byte MFMA operand.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 as_bf16x8(f32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {          // v_cvt_pk_bf16_f32, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void split_bf16(f32x4 v0, f32x4 v1, f32x4& heads, f32x4& tails) {
  const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned h = pack_bf16(v[2 * p], v[2 * p + 1]);
    const float t0 = v[2 * p] - __uint_as_float(h << 16), t1 = v[2 * p + 1] - __uint_as_float(h & 0xffff0000u);   // exact
    heads[p] = __uint_as_float(h);
    tails[p] = __uint_as_float(pack_bf16(t0, t1));
  }
}

// sum over the 32 lanes of a half-wave (all of them end up with it): four DPP adds inside each row of 16 lanes (quad
// permutes, half-row and row mirrors) and one cross-row exchange
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror
  v = dpp_add<0x140>(v);     // row_mirror
  return v + __shfl_xor(v, 16, 64);
}

// LDS tiles are k-interleaved: element (k, x) of a K-step lives at [(k/4)][x][k%4], so that
//   * the packed weights (stored the same way in HBM, see pack_weights) move global->LDS as 16-byte words,
//   * a thread that gathered 4 consecutive k rows of one pixel stores them with one ds_write_b128,
//   * one ds_read_b128 per operand feeds FOUR MFMAs: lane (i, h) reads k = 8g + 4h + {0..3} and MFMA
//     j of the group contracts the k pair {8g + j, 8g + 4 + j} (any pairing is valid as long as the A
//     and B operands agree) -- 8 LDS reads per 32 MFMAs instead of 32.
// FAST: the gathered tensor's channel count is a multiple of BK, so a K-step never straddles two
// taps: one (dh, dw) per step, validity and the spatial offset computed once per thread and step, the
// channel offset of each row rides in the scalar operand (zero VALU per gathered element).
// STREAMK: persistent launch (3 workers per CU).  The (tile, K-step) iteration space is cut into equal
// contiguous ranges, one per worker, so every CU gets the same amount of matrix work no matter how the
// tile count divides by the CU count (1178 tiles on 256 CUs would otherwise run 5 "rounds" for 4.6 of
// work).  A tile cut by a range boundary is finished by the worker holding its FIRST K-steps (it reaches
// them last); the other worker deposits its accumulators in `partial` as soon as it has them -- the tile segment a
// range starts with is the first thing it computes, before it waits for anything itself, so no co-residency of all
// workers is required -- and raises a flag.  The hand-off is placement independent (the per-XCD L2s are not coherent):
// deposits are written with agent-scope (sc1, write-through) 16-byte stores and read with sc1 loads, the flag is an
// agent-scope atomic that its single consumer resets (self-cleaning: no memset between launches).
// BITS: 0 = fp32 epilogue operands only; 1 = the ReLU epilogue also records its pattern as bits (Epilogue::obits);
//       2 = the epilogue masks with a recorded bit pattern (Epilogue::mbits).  Separate instantiations, so that the plain
//       kernels carry none of the extra scalar state.
// SK: 0 = one block per tile; 1 = persistent stream-K (above); 2 = one block per tile for the leading whole rounds of resident
//     blocks AND, in the same launch, the remaining `tail_tiles` tiles cut into `tail_split` K-ranges of one block each (round 6).
//     A cut tile's pieces with the LATER K-steps carry the lower block ids -- blocks are dispatched in id order, so they are resident
//     or done before the piece with the first K-steps (the owner: it adds their deposits and runs the epilogue) even starts, and
//     they wait for nothing: progress needs no co-residency.  Same sc1 deposits and self-cleaning flags as the persistent kernel,
//     but at the plain kernel's four blocks per CU (a piece is ONE tile segment: no loop over segments, deposits read a quarter
//     at a time) and without a second launch: the tail's ramp hides behind the last full round's stragglers.
template <int BM, int BN, int WAVES_M, int BK, bool FAST, int SK, bool X3, int BITS = 0>
__global__ __launch_bounds__(kThreads, SK == 1 ? kSkWorkersPerCu : 4) void conv_gemm(const float* __restrict__ X, const float* __restrict__ Wp,
                                                      const int4* __restrict__ tab, float* __restrict__ Out,
                                                      GemmGeom g, Epilogue ep, int m_tiles, int n_tiles,
                                                      float* __restrict__ partial, int* __restrict__ flags, int tail_tiles,
                                                      int tail_split) {
  constexpr bool STREAMK = SK == 1;      // persistent: a worker loops over tile segments
  constexpr bool TAIL = SK == 2;
  constexpr bool HANDOFF = STREAMK || TAIL;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int KQ = BK / 4;                                // k quads per K-step
  constexpr int A_VEC = KQ * BM;                             // 16-byte words per weight tile
  constexpr int A_PER_T = (A_VEC + kThreads - 1) / kThreads;
  constexpr int B_Q_PASS = kThreads / BN > 0 ? kThreads / BN : 1;   // k quads covered by one pass of the block
  constexpr int B_QUADS = KQ / B_Q_PASS;                     // quads (of 4 rows) gathered per thread
  constexpr int ACC_REGS = TM * TN * 16;
  static_assert(TM >= 1 && TN >= 1 && BN <= kThreads && KQ % B_Q_PASS == 0, "tile shape");
  static_assert(!X3 || (BK == 16 && BN == 128), "split-bf16 path: one 16-deep MFMA block per K-step, a k octet per thread");

  __shared__ f32x4 sA[2][KQ * BM];
  __shared__ f32x4 sB[2][KQ * BN];
  // Epilogue operands that are per ROW (shift) or per row and 32-pixel group (mask words) wait in LDS from the tile's prologue
  // on: the epilogue then contains no vector-memory load except the residual's.  That matters because gfx9's vmcnt retires
  // loads AND stores in issue order -- a wait for any load issued after a store is also a wait for that store's
  // acknowledgement; rounds 1-4 interleaved 8 batches of (loads, 8 stores) and paid 8 store round trips per tile
  // (28-36 us of a 91 us workgroup life on the K = 256 layers, profiles/r3_tile_timeline.txt).
  __shared__ __attribute__((aligned(16))) float s_shift[BM];
  // byte offset of (image, oh, ow) of each of the tile's BN pixel columns inside one output row, or the poison offset: formed in
  // the prologue next to the gather's own pixel decode (the same two integer divisions), read back by the epilogue -- whose
  // instructions are only served where the co-resident K-loop waves stall, ~20 per microsecond (profiles/r5_epilogue_anatomy.txt)
  __shared__ unsigned s_vo[BN];
  __shared__ __attribute__((aligned(16))) unsigned s_mbits[BITS == 2 ? (BN / 32) * BM : 4];   // [32-pixel group][row]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform on purpose: scalar buffer offsets
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, g.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, g.w_bytes);

  const int OHW = g.OH * g.OW;
  const int planeHW = g.H * g.W;
  const int bcol = t % BN;
  const int bq0 = __builtin_amdgcn_readfirstlane(t / BN);    // wave-uniform first quad
  const int a_step_bytes = KQ * g.Mpad * 16;
  const int KT = g.Kpad / BK;
  const int OutHW = g.OutH * g.OutW;
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(Out, g.out_bytes);
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(ep.res ? ep.res : Out, g.out_bytes);
  const __amdgpu_buffer_rsrc_t rmsk = BITS == 2 ? make_rsrc(ep.mbits, g.M * ep.w32 * 4) : make_rsrc(ep.mask ? ep.mask : Out, g.out_bytes);
  const __amdgpu_buffer_rsrc_t rsh = make_rsrc(ep.shift ? ep.shift : Out, ep.shift ? g.M * 4 : 0);
  const bool ragged = (g.M & 7) != 0;
  const __amdgpu_buffer_rsrc_t rpart = make_rsrc(HANDOFF ? (const void*)partial : (const void*)Out,
                                                 HANDOFF ? (STREAMK ? kSkWorkers : kTailSlots) * ACC_REGS * kThreads * 4 : 0);

  // ---- which part of the iteration space is mine --------------------------------------------
  // block b runs on XCD b%8: tiles (or ranges) that are adjacent -- same activation tile, next M tile --
  // go to the same XCD so that they share its L2.
  const int bid = blockIdx.x;
  const int xcd = bid % kNumXcd, slot = bid / kNumXcd;
  int it, it_end;                                            // (tile, K-step) space: tiles*KT < 2^31 (checked by the host)
  int my_range = 0;
  // stream-K ranges: G workers (= gridDim.x: 3 per CU on all CUs, or on fewer when CUs are left to overlapped collectives,
  // dasac_set_reserved_cus) split the (tile, K-step) space into contiguous ranges of sk_per or sk_per + 1 steps
  // (the first sk_rem ranges get the extra one): range r starts at r * sk_per + min(r, sk_rem)
  int sk_per = 0, sk_rem = 0;
  auto sk_start = [&](int r) { return r * sk_per + min(r, sk_rem); };
  if (STREAMK) {
    const int G = gridDim.x;
    my_range = xcd * (G / kNumXcd) + slot;
    const int sk_total = m_tiles * n_tiles * KT;             // < 2^31 (checked by the host)
    sk_per = sk_total / G;
    sk_rem = sk_total - sk_per * G;
    it = sk_start(my_range);
    it_end = sk_start(my_range + 1);
  } else {
    // each XCD owns a CONTIGUOUS run of pixel tiles (spatial neighbours share halo rows in its L2)
    const int per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
    const int lead_blocks = per_xcd * kNumXcd * m_tiles;       // == gridDim.x without a tail
    if (!TAIL || bid < lead_blocks) {
      const int n_local = slot / m_tiles;
      const int n_tile = xcd * per_xcd + n_local;
      if (n_local >= per_xcd || n_tile >= n_tiles) return;
      it = (n_tile * m_tiles + slot % m_tiles) * KT;
      it_end = it + KT;
    } else {
      // tail piece q: K-range `part` (later ranges first) of tail tile tr; the tail tiles follow the n_tiles leading pixel tiles;
      // runs of adjacent tiles (the M tiles of a pixel tile, neighbouring pixel tiles) go to one XCD, like the leading part
      const int q = bid - lead_blocks;
      const int per_x = (tail_tiles + kNumXcd - 1) / kNumXcd;
      const int part_rev = q / (per_x * kNumXcd), ql = q - part_rev * (per_x * kNumXcd);
      const int tl = ql / kNumXcd;                             // ql % kNumXcd == xcd (lead_blocks and per_x * kNumXcd are multiples of 8)
      const int tr = xcd * per_x + tl;
      if (tr >= tail_tiles) return;
      const int part = tail_split - 1 - part_rev;
      const int tile = n_tiles * m_tiles + tr;
      // balanced K-ranges: part p holds steps [p*KT/s, (p+1)*KT/s)
      it = tile * KT + part * KT / tail_split;                 // (KT <= 2^15 or so: no overflow)
      it_end = tile * KT + (part + 1) * KT / tail_split;
      my_range = tr * (tail_split - 1) + part - 1;             // deposit / flag slot of a non-owner piece (part >= 1)
    }
  }

  // (a do-while whose back edge exists only in the persistent variant: the tile-per-block kernel runs the body once -- it returned
  // above if it has no tile --, and the compiler must SEE that, or every K-loop invariant stays live across the epilogue
  // and its register pressure pushes them into spill slots that are then re-read inside the K loop)
  do {
    const int tile = it / KT;
    const int ks = it - tile * KT;
    const int ke = min(KT, ks + (it_end - it));
    it += ke - ks;
    const int m0 = (tile % m_tiles) * BM, n0 = (g.n_tile0 + tile / m_tiles) * BN;

    // ---- this thread's pixel column of the gather -------------------------------------------
    int pixbase, ih0, iw0;
    unsigned pro_vo = kPoison;
    {
      const int pix = n0 + bcol;
      if (pix < g.Npix) {
        const int n = pix / OHW, r = pix - n * OHW;
        const int oh = r / g.OW, ow = r - oh * g.OW;
        ih0 = oh * g.stride;
        iw0 = ow * g.stride;
        pixbase = n * g.CxHW + ih0 * g.W + iw0;
        pro_vo = (unsigned)(n * g.M * OutHW + oh * g.ostride * g.OutW + ow * g.ostride) * 4u;
      } else {
        ih0 = -kInvalid;     // every tap out of bounds -> poison offsets -> zeros
        iw0 = 0;
        pixbase = 0;
      }
    }
    // weight tile: per-thread byte offsets inside a K-step, the K-step itself goes in the scalar offset
    unsigned voff_a[A_PER_T];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int v = t + i * kThreads;
      const int q = v / BM, m = v - q * BM;
      voff_a[i] = (unsigned)(q * g.Mpad + m0 + m) * 16u;
    }

    f32x4 ra[A_PER_T];
    f32x4 rb[B_QUADS];

#define DASAC_LOAD_TILE(kt)                                                                          \
  {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < A_PER_T; ++i) {                                            \
      if (A_VEC % kThreads == 0 || t + i * kThreads < A_VEC) ra[i] = buf_f32x4(rw, voff_a[i], (kt) * a_step_bytes); \
    }                                                                                                \
    if (FAST) {                                                                                      \
      const int4 e = tab[(kt) * BK]; /* one tap per K-step: (koff, dh, dw, channel offset) */        \
      const int ih = ih0 + e.y, iw = iw0 + e.z;                                                      \
      const bool ok = ((unsigned)ih < (unsigned)g.H) & ((unsigned)iw < (unsigned)g.W);               \
      /* the K-step's tap and first channel plane ride in the per-lane offset (one VALU add); what is left for the */ \
      /* scalar operand -- the row's plane within the step -- is loop invariant: no scalar arithmetic per load      */ \
      const unsigned voff = ok ? (unsigned)(pixbase + e.x) * 4u : kPoison;                           \
      _Pragma("unroll") for (int r = 0; r < B_QUADS; ++r) {                                          \
        const int row0 = X3 ? 8 * bq0 + 4 * r : 4 * (bq0 + r * B_Q_PASS);                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
          rb[r][j] = buf_f32(rx, voff, (row0 + j) * planeHW * 4);                                    \
      }                                                                                              \
    } else {                                                                                         \
      int4 te[B_QUADS * 4]; /* all table rows of this K-step first: one batch of scalar loads */     \
      _Pragma("unroll") for (int r = 0; r < B_QUADS; ++r)                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
          te[r * 4 + j] = tab[(kt) * BK + (X3 ? 8 * bq0 + 4 * r : 4 * (bq0 + r * B_Q_PASS)) + j];    \
      _Pragma("unroll") for (int r = 0; r < B_QUADS; ++r) {                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
          const int4 e = te[r * 4 + j];                                                              \
          const int ih = ih0 + e.y, iw = iw0 + e.z;                                                  \
          const bool ok = ((unsigned)ih < (unsigned)g.H) & ((unsigned)iw < (unsigned)g.W);           \
          rb[r][j] = buf_f32(rx, ok ? (unsigned)(pixbase + e.x) * 4u : kPoison, 0);                  \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }
#define DASAC_STORE_TILE(buf)                                                                        \
  {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < A_PER_T; ++i) {                                            \
      const int v = t + i * kThreads;                                                                \
      if (A_VEC % kThreads == 0 || v < A_VEC) sA[buf][v] = ra[i];                                    \
    }                                                                                                \
    if (X3) {                                                                                        \
      f32x4 hi, lo;                                                                                  \
      split_bf16(rb[0], rb[B_QUADS - 1], hi, lo);                                                    \
      sB[buf][(2 * bq0) * BN + bcol] = hi;                                                           \
      sB[buf][(2 * bq0 + 1) * BN + bcol] = lo;                                                       \
    } else {                                                                                         \
      _Pragma("unroll") for (int r = 0; r < B_QUADS; ++r) sB[buf][(bq0 + r * B_Q_PASS) * BN + bcol] = rb[r]; \
    }                                                                                                \
  }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    DASAC_STAMP(0);
    // the tile's per-row epilogue operands (consumed only by the worker that runs the epilogue: ks == 0)
    float pro_shift = 0.f;
    unsigned pro_bits[BITS == 2 ? (BN / 32) * BM / kThreads : 1];
    if (!HANDOFF || ks == 0) {
      if (t < BM) pro_shift = buf_f32(rsh, (unsigned)(m0 + t) * 4u, 0);      // rows past M (and a null shift: 0 records) read 0
      if constexpr (BITS == 2) {
        static_assert(BITS != 2 || ((BN / 32) * BM) % kThreads == 0, "mask words per tile");
#pragma unroll
        for (int u = 0; u < (BN / 32) * BM / kThreads; ++u) {
          const int idx = t + u * kThreads, c = idx / BM, row = idx - c * BM;
          const int wcol = (n0 >> 5) + c;
          pro_bits[u] = __builtin_amdgcn_raw_buffer_load_b32(
              rmsk, (m0 + row < g.M && wcol < ep.w32) ? (unsigned)((m0 + row) * ep.w32 + wcol) * 4u : kPoison, 0, 0);
        }
      }
    }
    DASAC_LOAD_TILE(ks);
    DASAC_STORE_TILE(ks & 1);
    if (!HANDOFF || ks == 0) {
      if (t < BM) s_shift[t] = pro_shift;
      if (t < BN) s_vo[t] = pro_vo;
      if constexpr (BITS == 2) {
#pragma unroll
        for (int u = 0; u < (BN / 32) * BM / kThreads; ++u) s_mbits[t + u * kThreads] = pro_bits[u];
      }
    }
    __syncthreads();
    DASAC_STAMP(1);
    for (int kt = ks; kt < ke; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < ke;
      if (more) DASAC_LOAD_TILE(kt + 1);
      f32x4 a4[BK / 8][TM], b4[BK / 8][TN];
      if (X3) {
        // quad slot q of the tile = (k octet lh, half h): h = 0 the bf16 heads, h = 1 the bf16 tails
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a4[h][i] = sA[buf][(2 * lh + h) * BM + wm * WM + i * 32 + li];
#pragma unroll
          for (int j = 0; j < TN; ++j) b4[h][j] = sB[buf][(2 * lh + h) * BN + wn * WN + j * 32 + li];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            // x*y ~ xl*yh + xh*yl + xh*yh (the tail*tail term is below 2^-16 of the product), fp32 accumulate
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a4[1][i]), as_bf16x8(b4[0][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a4[0][i]), as_bf16x8(b4[1][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a4[0][i]), as_bf16x8(b4[0][j]), acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
      for (int gq = 0; gq < BK / 8; ++gq) {
        const int q = 2 * gq + lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) a4[gq][i] = sA[buf][q * BM + wm * WM + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) b4[gq][j] = sB[buf][q * BN + wn * WN + j * 32 + li];
      }
#pragma unroll
      for (int gq = 0; gq < BK / 8; ++gq) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[gq][i].x, b4[gq][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[gq][i].y, b4[gq][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[gq][i].z, b4[gq][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[gq][i].w, b4[gq][j].w, acc[i][j], 0, 0, 0);
          }
      }
      }
      if (more) DASAC_STORE_TILE(buf ^ 1);
      __syncthreads();
    }
#undef DASAC_LOAD_TILE
#undef DASAC_STORE_TILE

    if (HANDOFF) {
      // Hand-off of a tile cut by range boundaries: the worker holding a tile's LATER K-steps deposits its accumulators in
      // `partial` and raises a flag, the worker holding the FIRST K-steps (the owner) adds the deposits and runs the epilogue.
      // Deposits are agent-scope (sc1) write-through stores read back with sc1 loads -- placement independent (the per-XCD
      // L2s are not coherent with each other), no buffer_wbl2 / buffer_inv fences.  (Measured in round 3: routing the
      // deposits of same-XCD neighbours through L2 with plain accesses instead -- whole tiles per XCD, hardware XCC id
      // checked -- changes nothing: 116.2 vs 113.2 us on the layer3 3x3 remainder.  The ~25 us a remainder launch costs
      // beyond its K-steps is ramp, hand-off latency and epilogue, not the 2 x 50 MB of deposit traffic.)
      constexpr int kAux = kAuxSc1;
      int c_first = 0, c_count = 0;
      if (ks > 0) {
        // partial layout [range][32x32 sub-tile][quarter][thread][4]: 16-byte stores, a wave writes 1 KB contiguous
        const int dst0 = my_range * (ACC_REGS * kThreads * 4);     // bytes; scalar
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
              __builtin_amdgcn_raw_buffer_store_b128(
                  u32x4{__float_as_uint(acc[i][j][4 * r4]), __float_as_uint(acc[i][j][4 * r4 + 1]), __float_as_uint(acc[i][j][4 * r4 + 2]),
                        __float_as_uint(acc[i][j][4 * r4 + 3])},
                  rpart, (unsigned)t * 16u, dst0 + ((i * TN + j) * 4 + r4) * kThreads * 16, kAux);
            asm volatile("" ::: "memory");
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's stores are acknowledged (L2 / memory)
        __syncthreads();
        if (t == 0) __hip_atomic_store(&flags[my_range], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (ke < KT) {
        // I hold the FIRST K-steps; the rest was deposited by the following range(s), each at the very start of its work
        // (a range shorter than a tile -- fewer tiles than workers -- makes several of them contribute).  Ranges are never
        // empty (the host picks this schedule only with >= 1 K-step per worker).
        const int tile_end = (tile + 1) * KT;
        int n_contrib = 1;                                     // contributors are my_range + 1 .. my_range + n_contrib
        if (TAIL) {                                            // the tile's other pieces: slots tr * (s - 1) + 0 .. s - 2
          my_range = (tile - n_tiles * m_tiles) * (tail_split - 1) - 1;
          n_contrib = tail_split - 1;
        } else {
          while (sk_start(my_range + n_contrib + 1) < tile_end) ++n_contrib;
        }
        if (t == 0) {
          // Progress does not need all workers resident: a range deposits at the very START of its work, before it waits for
          // anything, and blocks are dispatched in id order, so the depositor of r is at worst the next block to get a slot.
          // A wait that still outlasts ~2^26 sleeps (seconds) means the depositor died: abort the kernel (the stream reports
          // a launch failure) instead of summing a slot that was never written.  ALL flags are collected before the first
          // load: the depositors reach their boundaries together, so this costs one wait, not one per contributor.
          for (int r = my_range + 1; r <= my_range + n_contrib; ++r) {
            int spins = 0;
            while (__hip_atomic_load(&flags[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins >= (1 << 26)) __builtin_trap();
            }
            // self-cleaning flags: a range deposits at most once per launch and this worker is its only consumer, so the
            // flag can go back to 0 right here -- the next launch on the stream finds the array zeroed (no memset per conv)
            __hip_atomic_store(&flags[r], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __syncthreads();
        c_first = my_range + 1;
        c_count = n_contrib;
      }
      // the owner adds its contributors' deposits (a loop that runs zero times for everybody else: written outside the branches
      // above so that the accumulators are loop-carried in place instead of being merged -- copied, at 128 registers spilled --
      // at the join of three paths)
      {
        for (int r = c_first; r < c_first + c_count; ++r) {
          const int src0 = r * (ACC_REGS * kThreads * 4);
          // half a deposit in flight at a time: 8 x 16-byte loads per thread (a whole one -- 64 more registers next to the 64
          // accumulators -- does not fit the 168 registers of 3 workers per CU without spilling); a quarter in the tail pieces'
          // 128-register kernel
          constexpr int NB = TAIL ? 4 : 2;
          constexpr int QH = TM * TN * 4 / NB;
#pragma unroll
          for (int hb = 0; hb < NB; ++hb) {
            f32x4 pv[QH];
#pragma unroll
            for (int q = 0; q < QH; ++q) {
              const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rpart, (unsigned)t * 16u, src0 + (hb * QH + q) * kThreads * 16, kAux);
              pv[q] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            }
#pragma unroll
            for (int q = 0; q < QH; ++q) {
              const int qq = hb * QH + q, sub = qq >> 2, r4 = qq & 3;
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[sub / TN][sub % TN][4 * r4 + e] += pv[q][e];
            }
            asm volatile("" ::: "memory");
          }
        }
      }
    }

    // ---- epilogue: (+shift | bias) (+residual) (ReLU) (ReLU-backward mask) -> store ----------------
    // The BN scale is already folded into the packed weights.  All traffic goes through buffer
    // descriptors: per-lane voffset = pixel position (+ the lane-half's 4-row step), the row offset is
    // scalar; loads of a 16-row group are issued as one batch before any of them is consumed.
    const bool deposited = HANDOFF && ks > 0;                // this worker only contributed a partial sum
    DASAC_STAMP(2);
    // (Round 5, measured: __builtin_amdgcn_s_setprio(3) for the epilogue changes nothing -- 21.0 vs 21.3 us from the end of the K loop to
    // the last store issued, profiles/r5_epilogue_anatomy.txt: a wave with an MFMA ready is served first whatever the priorities are.
    // An s_nop 15 after every (second, fourth) MFMA of the short-K launches, to open issue windows for co-resident prologue /
    // epilogue waves, buys 1-2.5 % on the K = 256 layers and costs 0.7 % at K = 4608: not kept.)
    if (!deposited) {
    if constexpr (BITS != 3) {
    // Per 32-pixel column group j: ONE batch of loads (the residual and / or an fp32 mask, TM x 16 values; shift and mask words
    // come from LDS), then the arithmetic and the TM x 16 stores.  A layer without residual issues no vector-memory load here
    // at all -- 64 stores back to back, the workgroup ends one acknowledgement after the last; with a residual the second
    // batch's wait covers the first batch's stores: two store round trips per tile instead of eight.
    // FULL: the tile's BM rows all exist (every tile of a layer whose channel count is a multiple of BM): no row tests.
    // (the plane size is laundered through an empty asm so that none of the 64 per-row scalar offsets derived from it is
    // computed -- and kept alive in SGPRs, then spilled into VGPR lanes -- ahead of the K loop)
    int OutHWe = OutHW;
    asm volatile("" : "+s"(OutHWe));
    auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;
      // row of accumulator register rg of row group i: mrow + (rg&3) + 8*(rg>>2), + 4*lh in the lane offset -> a row exists
      // iff its lane-independent part is below M - 4*lh (one per-lane limit, a compare and a select per access)
      const int mlim = g.M - 4 * lh;
      const unsigned lane_bit = 1u << li;                      // this lane's pixel inside a mask word
#define DASAC_ROW(i, rg) (m0 + wm * WM + (i) * 32 + ((rg) & 3) + 8 * ((rg) >> 2))
#define DASAC_VOFF(i, rg) (FULL ? vo : (DASAC_ROW(i, rg) < mlim ? vo : kPoison))
      // one batch: row groups I0 .. I0+NI-1 of column group j
      auto batch = [&](int j, unsigned vo, auto i0_tag, auto ni_tag, auto relu_tag) __attribute__((always_inline)) {
        constexpr int I0 = decltype(i0_tag)::value, NI = decltype(ni_tag)::value;
        constexpr bool RELU = decltype(relu_tag)::value;     // compile time: a run-time flag costs a v_max AND a select per element
        const int wcol = (n0 + wn * WN + j * 32) >> 5;          // this wave's 32-pixel group = one word column of the bit masks
        float rs[NI][16], mk[NI][16];
        if (ep.res) {
#pragma unroll
          for (int i = I0; i < I0 + NI; ++i)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) rs[i - I0][rg] = buf_f32(rres, DASAC_VOFF(i, rg), DASAC_ROW(i, rg) * OutHWe * 4);
        }
        if (BITS != 2 && ep.mask) {          // fp32 mask: the ReLU-backward pattern read from the activation itself
#pragma unroll
          for (int i = I0; i < I0 + NI; ++i)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) mk[i - I0][rg] = buf_f32(rmsk, DASAC_VOFF(i, rg), DASAC_ROW(i, rg) * OutHWe * 4);
        }
#pragma unroll
        for (int i = I0; i < I0 + NI; ++i) {
          const int mrow = m0 + wm * WM + i * 32;
          if (!FULL && mrow >= g.M) continue;                  // whole 32-row group beyond M (uniform)
          const int lrow = wm * WM + i * 32 + 4 * lh;            // + 8*q + e: the four rows of accumulator registers 4q .. 4q+3
          int bitrows = 0;                                     // producer: lane r collects the mask word of row mrow + r
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(&s_shift[lrow + 8 * q]);
            u32x4 mb4 = u32x4{0u, 0u, 0u, 0u};
            if constexpr (BITS == 2) mb4 = *reinterpret_cast<const u32x4*>(&s_mbits[(wn * (WN / 32) + j) * BM + lrow + 8 * q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int rg = 4 * q + e;
              float v = acc[i][j][rg] + sh4[e];
              if (ep.res) v = v + rs[i - I0][rg];
              if constexpr (RELU) v = fmaxf(v, 0.f);
              if (BITS != 2 && ep.mask) v = mk[i - I0][rg] > 0.f ? v : 0.f;
              if constexpr (BITS == 2) v = (mb4[e] & lane_bit) ? v : 0.f;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, DASAC_VOFF(i, rg), DASAC_ROW(i, rg) * OutHWe * 4, 0);
              if constexpr (BITS == 1)      // lanes 0-31 hold row (rg&3) + 8*(rg>>2), lanes 32-63 the row 4 below it: two words per ballot
                bitrows = put_mask_rows(rg, __builtin_amdgcn_ballot_w64(v > 0.f), bitrows);
            }
          }
          if constexpr (BITS == 1) {
            // (a 32-pixel group of the last tile may lie entirely past the last pixel: no word exists for it)
            if (lane < 32 && mrow + lane < g.M && wcol < ep.w32) ep.obits[(size_t)(mrow + lane) * ep.w32 + wcol] = (unsigned)bitrows;
          }
        }
        asm volatile("" ::: "memory");     // the next batch's loads stay behind this batch's stores (registers -> occupancy)
      };
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        unsigned vo = s_vo[wn * WN + j * 32 + li];
        vo = vo == kPoison ? kPoison : vo + (unsigned)(4 * lh * OutHWe) * 4u;      // + the lane half's four rows
        using c0 = std::integral_constant<int, 0>;
        using c1 = std::integral_constant<int, 1>;
        auto run = [&](auto relu_tag) __attribute__((always_inline)) {
          if (TM == 2 && BITS != 2 && ep.res && ep.mask) {   // residual AND fp32 mask: 2 x 16 loaded values per row group -> one group per batch
            batch(j, vo, c0{}, c1{}, relu_tag);
            if constexpr (TM == 2) batch(j, vo, c1{}, c1{}, relu_tag);
          } else {
            batch(j, vo, c0{}, std::integral_constant<int, TM>{}, relu_tag);
          }
        };
        if (ep.relu) run(std::true_type{});
        else run(std::false_type{});
      }
#undef DASAC_ROW
#undef DASAC_VOFF
    };
    if (m0 + BM <= g.M) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    } else {
      // ---- BITS == 3: raw convolution (+ bias) in front of a batch-statistics BatchNorm.  No residual / ReLU / mask here (they
      // follow the normalisation); instead the per-row sum and sum of squares of the stored values.  Row groups are the OUTER
      // loop so that only one group's 2 x 16 running sums are live next to the accumulators.  Lane (li, lh) holds, per
      // accumulator register rg, the sum over ITS pixels of row i*32 + (rg&3) + 8*(rg>>2) + 4*lh: sum over the 32 lanes of
      // the half-wave, lane li == rg keeps row rg's total, the WAVES_N waves of a row meet in LDS (the operand tiles are
      // dead: the K loop ended with a barrier), 2*BM coalesced floats go out per tile.
      static_assert(BITS != 3 || (BM == 128 && WAVES_N == 2 && TN == 2), "stats epilogue: 128-row tile, two pixel waves");
      float* sst = reinterpret_cast<float*>(&sA[0][0]);        // [2][BM][WAVES_N]
      unsigned vo[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int pix = n0 + wn * WN + j * 32 + li;
        vo[j] = kPoison;
        if (pix < g.Npix) {
          const int n = pix / OHW, r = pix - n * OHW;
          const int oh = r / g.OW, ow = r - oh * g.OW;
          vo[j] = (unsigned)(n * g.M * OutHW + oh * g.ostride * g.OutW + ow * g.ostride + 4 * lh * OutHW) * 4u;
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mrow = m0 + wm * WM + i * 32;
        float ms = 0.f, mq = 0.f;
        if (mrow < g.M) {                                        // (uniform) a whole 32-row group beyond M contributes zeros
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            float sh[8];
            bool rowok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int rg = half * 8 + u;
              const int mr = mrow + (rg & 3) + 8 * (rg >> 2);
              rowok[u] = ragged ? (mr + 4 * lh < g.M) : (mr < g.M);
              sh[u] = s_shift[wm * WM + i * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * lh];    // 0 for rows past M / no bias
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int rg = half * 8 + u;
              const int roff = (mrow + (rg & 3) + 8 * (rg >> 2)) * OutHW * 4;
              float ss = 0.f, qq = 0.f;
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const float v = acc[i][j][rg] + sh[u];
                const unsigned vr = rowok[u] ? vo[j] : kPoison;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, vr, roff, 0);
                const float vv = vr != kPoison ? v : 0.f;        // pixels / rows past the tensor count as 0
                ss += vv;
                qq = __builtin_fmaf(vv, vv, qq);
              }
              const float rs_ = half_wave_sum(ss), rq_ = half_wave_sum(qq);
              ms = li == rg ? rs_ : ms;
              mq = li == rg ? rq_ : mq;
            }
            asm volatile("" ::: "memory");
          }
        }
        if (li < 16) {
          const int row = wm * WM + i * 32 + (li & 3) + 8 * (li >> 2) + 4 * lh;
          sst[row * WAVES_N + wn] = ms;
          sst[(BM + row) * WAVES_N + wn] = mq;
        }
      }
      __syncthreads();
      {
        const int which = t / BM, row = t - which * BM;          // 256 threads = {sum, sum of squares} x 128 rows
        const float tot = sst[(which * BM + row) * WAVES_N] + sst[(which * BM + row) * WAVES_N + 1];
        ep.stats[((size_t)(n0 / BN) * 2 + which) * g.Mpad + m0 + row] = tot;
      }
    }
    }
#ifdef DASAC_TRACE_TILES
    DASAC_STAMP(5);                                    // the epilogue's stores have been issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and acknowledged
    DASAC_STAMP(3);
    if (g_tile_trace && threadIdx.x == 0) g_tile_trace[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
#endif
    if (STREAMK) __syncthreads();      // LDS is reused by the next tile of this worker
  } while (STREAMK && it < it_end);
}

// ------------------------------------------------------------------------------------------
// weight-gradient kernel: reduction over pixels, split across blocks
//   P[split][m][k] = sum_{pix in chunk} dZ[m][pix] * gather(X,k,pix)
// LDS tiles keep the natural [row][pix] order with an odd row pitch (33) so that both the
// coalesced tile writes and the strided MFMA operand reads are bank-conflict free.
// ------------------------------------------------------------------------------------------
constexpr int kWgPix = 32;   // pixels per K-step
constexpr int kWgPitch = 36;   // row pitch (floats): 16-byte aligned rows, 16 consecutive rows x 4 dwords hit 64 distinct banks

// FAST: Cx % BN == 0 (the 128 im2col rows of a block belong to ONE tap) and M % 8 == 0: one (dh, dw)
// per block, the per-row part of both operand addresses is a scalar offset, the pixel position is
// advanced incrementally (no divisions in the loop) -> ~25 VALU per 64 MFMAs.
// X3: split-bf16 arithmetic (see conv_gemm).  The loader is unchanged (lane = pixel); every element is split
// into bf16 head + tail on its way to LDS (three VALU each: v_cvt_pk_bf16_f32 with the value in the HIGH half
// gives the head as an fp32 bit pattern directly) and stored with 16-bit writes into [octet of 8 pixels][row][8]
// operand words; the octet pitch rows+2 keeps the 64 lanes of a store on distinct banks and the 16-byte MFMA
// operand reads of consecutive rows contiguous.
template <int BM, int BN, int WAVES_M, bool FAST, bool X3, bool QUAD = false, bool QTAP = false>
__global__ __launch_bounds__(kThreads, 3) void conv_wgrad(const float* __restrict__ dZ, const float* __restrict__ X,
                                                       const int4* __restrict__ tab, float* __restrict__ P,
                                                       float* __restrict__ Psum, GemmGeom g, int m_tiles, int k_tiles,
                                                       int n_splits, int pix_per_split) {
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_LOADS = BM / 8, B_LOADS = BN / 8;      // 8 rows of 32 pixels per pass of 256 threads

  // ONE LDS stage (34 KB for the 128x128 tile -> 4 blocks per CU); the next tile waits in registers while the
  // MFMAs run, two barriers per step.  More co-resident blocks hide the extra barrier (measured).
  constexpr int PA = BM + 2, PB = BN + 2;                // X3: operand words per pixel octet (rows + 2)
  __shared__ __attribute__((aligned(16))) float sA[1][X3 ? 32 * PA : BM * kWgPitch];   // X3: [head|tail][4 octets][PA] x 16 B
  __shared__ __attribute__((aligned(16))) float sB[1][X3 ? 32 * PB : BN * kWgPitch];

  // block b runs on XCD b % 8: give every XCD a CONTIGUOUS run of the (split-major) block list, so that the tiles of a
  // pixel split -- which all stream the same dZ and X pixels -- meet in one L2 instead of being fetched by all eight
  const int n_blocks = m_tiles * k_tiles * n_splits;
  const int per_xcd = (n_blocks + kNumXcd - 1) / kNumXcd;
  const int lb = ((int)blockIdx.x % kNumXcd) * per_xcd + (int)blockIdx.x / kNumXcd;
  if (lb >= n_blocks) return;
  const int tile = lb % (m_tiles * k_tiles), split = lb / (m_tiles * k_tiles);
  const int m_tile = tile % m_tiles, k_tile = tile / m_tiles;
  const int m0 = m_tile * BM, kb0 = k_tile * BN;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform on purpose: scalar buffer offsets
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;
  const int pl = t & 31, prow = t >> 5;                  // loader: pixel lane, row within pass

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, g.x_bytes);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(dZ, g.z_bytes);

  const int OHW = g.OH * g.OW;
  const int planeHW = g.H * g.W;
  const int p_begin = split * pix_per_split;
  const int p_end = min(p_begin + pix_per_split, g.Npix);
  const int steps = (p_end - p_begin + kWgPix - 1) / kWgPix;

  // generic path: table rows of this thread's B loads are fixed for the whole kernel (offset, dh:dw packed)
  int te_off[FAST ? 1 : B_LOADS], te_d[FAST ? 1 : B_LOADS];
  int4 e0 = make_int4(0, 0, 0, 0);
  if (FAST) {
    e0 = tab[kb0];                                       // the block's tap: (koff, dh, dw, channel offset)
  } else {
#pragma unroll
    for (int r = 0; r < B_LOADS; ++r) {
      const int4 e = tab[kb0 + prow + r * 8];
      te_off[r] = e.x;
      te_d[r] = e.y >= kInvalid ? (int)0x80000000 : ((e.y << 16) | (e.z & 0xffff));   // dh = -32768 never in bounds
    }
  }

  const bool do_sums = (k_tile == 0) && (Psum != nullptr);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (QUAD) {
    // ---- 1x1 stride-1 convolutions (host: one tap at (0,0), H == OH, W == OW, M % 32 == 0): FOUR consecutive pixels per lane.
    // Both operands are plain [row][pixel] matrices there, so a lane's four pixels are 16 contiguous bytes: per 64 MFMAs a
    // thread issues 4+4 dwordx4 loads and 4+4 ds_write_b128 instead of 16+16 dword loads and 16+16 ds_write_b32.  Quads that
    // leave the split or the image (the dz / x tensors are [N][rows][OH*OW]: a quad may straddle two images) send their
    // whole wave through a per-element path for that step.
    static_assert(FAST && !X3 && BM % 32 == 0 && BN % 32 == 0, "quad loader: fp32, one tap per k tile, 32-row passes");
    static_assert(!QTAP || QUAD, "the tap variant extends the quad loader");
    // QTAP (round 6): the same loader for "same"-padded stride-1 k x k convolutions (host: H == OH, W == OW >= 4, one tap per k
    // tile).  Tap (dh, dw) reads x at flat position p + dh*W + dw of the SAME plane -- contiguous for a lane's four consecutive
    // pixels whatever row boundary lies between them -- so the gathered operand takes 16-byte loads too; what the tap changes is
    // VALIDITY (the convolution's zero padding), and that is per pixel: out-of-image taps are zeroed after the load.  Most quads
    // need nothing (one compare chain + a ballot); only waves that touch a row end or the first / last dh rows run the
    // per-element masks.  Earlier rounds kept these layers on the dword loader because "tap shifts break 16-byte runs at every
    // 97-pixel row": they break ALIGNMENT (dwordx4 needs 4 bytes) and validity, not contiguity.
    constexpr int AQ = BM / 32, BQ = BN / 32;
    const int pq = t & 7, prow4 = t >> 3;                  // loader: pixel quad of the step, row within a 32-row pass
    f32x4 qa[AQ], qb[BQ];
    float qsum[AQ];
#pragma unroll
    for (int i = 0; i < AQ; ++i) qsum[i] = 0.f;
    const int zimg_q = g.M * OHW;
    int pix = p_begin + 4 * pq;                            // first pixel of this thread's quad (advanced by kWgPix per step)
    int pn = pix / OHW, prr = pix - pn * OHW;              // image, flat position inside it
    int qoh = 0, qow = 0;                                  // QTAP: row / column of the quad's first pixel
    if (QTAP) {
      qoh = prr / g.OW;
      qow = prr - qoh * g.OW;
    }
    const int tap_off = QTAP ? e0.x - e0.w : 0;            // dh*W + dw
    // x offset of (image, k tile's first channel + prow4, quad start + tap), in elements.  The k tile's first channel rides in the
    // per-lane part so that it is negative only in the first rows of image 0, channel 0 (those quads take the per-element path:
    // a wrapped unsigned offset fails the range check whatever the scalar offset adds)
#define DASAC_WGQ_LOAD()                                                                             \
  {                                                                                                  \
    const int xo = pn * g.CxHW + (QTAP ? e0.w : 0) + prr + tap_off + prow4 * planeHW;                \
    const bool whole = (pix + 3 < p_end) & (prr + 3 < OHW) & (!QTAP || xo >= 0);                     \
    if (__builtin_amdgcn_ballot_w64(!whole) == 0) {                                                  \
      const unsigned vz = (unsigned)(pn * zimg_q + prr + prow4 * OHW) * 4u;                          \
      const unsigned vx = (unsigned)xo * 4u;                                                         \
      _Pragma("unroll") for (int i = 0; i < AQ; ++i) qa[i] = buf_f32x4(rz, vz, (m0 + i * 32) * OHW * 4); \
      _Pragma("unroll") for (int i = 0; i < BQ; ++i) qb[i] = buf_f32x4(rx, vx, ((QTAP ? 0 : e0.w) + i * 32 * planeHW) * 4); \
      if (QTAP) {                                                                                    \
        const int c0 = qow + e0.z;                                                                   \
        const bool inside = ((unsigned)(qoh + e0.y) < (unsigned)g.H) & (c0 >= 0) & (c0 + 3 < g.W) & (qow + 3 < g.OW); \
        if (__builtin_amdgcn_ballot_w64(!inside) != 0) {                                             \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
            int ow_e = qow + e, oh_e = qoh;                                                          \
            if (ow_e >= g.OW) { ow_e -= g.OW; ++oh_e; }                                              \
            const bool ok = ((unsigned)(oh_e + e0.y) < (unsigned)g.H) & ((unsigned)(ow_e + e0.z) < (unsigned)g.W); \
            _Pragma("unroll") for (int i = 0; i < BQ; ++i) qb[i][e] = ok ? qb[i][e] : 0.f;           \
          }                                                                                          \
        }                                                                                            \
      }                                                                                              \
    } else {                                                                                         \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
        int re = prr + e, ne = pn;                                                                   \
        while (re >= OHW) { re -= OHW; ++ne; } /* maps smaller than a quad: several images per quad */ \
        const bool ve = pix + e < p_end;                                                             \
        bool vt = ve;                                                                                \
        if (QTAP) {                                                                                  \
          const int oh_e = re / g.OW, ow_e = re - oh_e * g.OW;                                       \
          vt = ve & ((unsigned)(oh_e + e0.y) < (unsigned)g.H) & ((unsigned)(ow_e + e0.z) < (unsigned)g.W); \
        }                                                                                            \
        const unsigned vz = ve ? (unsigned)(ne * zimg_q + re + prow4 * OHW) * 4u : kPoison;          \
        const unsigned vx = vt ? (unsigned)(ne * g.CxHW + re + tap_off + prow4 * planeHW) * 4u : kPoison; \
        _Pragma("unroll") for (int i = 0; i < AQ; ++i) qa[i][e] = buf_f32(rz, vz, (m0 + i * 32) * OHW * 4); \
        _Pragma("unroll") for (int i = 0; i < BQ; ++i) qb[i][e] = buf_f32(rx, vx, (e0.w + i * 32 * planeHW) * 4); \
      }                                                                                              \
    }                                                                                                \
    if (do_sums) {                                                                                   \
      _Pragma("unroll") for (int i = 0; i < AQ; ++i) qsum[i] += (qa[i].x + qa[i].y) + (qa[i].z + qa[i].w); \
    }                                                                                                \
    pix += kWgPix;                                                                                   \
    prr += kWgPix;                                                                                   \
    if (QTAP) {                                                                                      \
      qow += kWgPix;                                                                                 \
      while (qow >= g.OW) { qow -= g.OW; ++qoh; }                                                    \
    }                                                                                                \
    while (prr >= OHW) { prr -= OHW; ++pn; if (QTAP) qoh -= g.OH; }                                  \
  }
#define DASAC_WGQ_STORE()                                                                            \
  {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < AQ; ++i)                                                   \
      *reinterpret_cast<f32x4*>(&sA[0][(prow4 + i * 32) * kWgPitch + 4 * pq]) = qa[i];               \
    _Pragma("unroll") for (int i = 0; i < BQ; ++i)                                                   \
      *reinterpret_cast<f32x4*>(&sB[0][(prow4 + i * 32) * kWgPitch + 4 * pq]) = qb[i];               \
  }
    if (steps > 0) {
      DASAC_WGQ_LOAD();
      DASAC_WGQ_STORE();
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      if (s + 1 < steps) DASAC_WGQ_LOAD();
      const float* a_base = &sA[0][(wm * WM + li) * kWgPitch + 4 * lh];
      const float* b_base = &sB[0][(wn * WN + li) * kWgPitch + 4 * lh];
#pragma unroll
      for (int gq = 0; gq < kWgPix / 8; ++gq) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * kWgPitch + 8 * gq);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * kWgPitch + 8 * gq);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      if (s + 1 < steps) {
        DASAC_WGQ_STORE();
        __syncthreads();
      }
    }
#undef DASAC_WGQ_LOAD
#undef DASAC_WGQ_STORE
    if (do_sums) {
      // the 8 lanes of a row hold its 32 pixels (4 each): butterfly over the three low lane bits
#pragma unroll
      for (int i = 0; i < AQ; ++i) {
        float v = qsum[i];
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (pq == 0) Psum[(size_t)split * g.Mpad + m0 + prow4 + i * 32] = v;
      }
    }
  } else {
  float ra[A_LOADS], rb[B_LOADS];
  float rsum[A_LOADS];                                   // per-row sums of dZ (only the k_tile 0 blocks)
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) rsum[i] = 0.f;
  const int zimg = g.M * OHW;                            // dZ image stride

  // pixel cursor of this thread (advanced by kWgPix per step)
  int pix = p_begin + pl;
  int pn = pix / OHW, prr = pix - pn * OHW;
  int poh = prr / g.OW, pow_ = prr - poh * g.OW;

#define DASAC_WG_LOAD()                                                                              \
  {                                                                                                  \
    const bool pv = pix < p_end;                                                                     \
    const int zbase = pn * zimg + poh * g.OW + pow_;                                                 \
    const int ih0 = poh * g.stride, iw0 = pow_ * g.stride;                                           \
    const int xbase = pn * g.CxHW + ih0 * g.W + iw0;                                                 \
    if (FAST) {                                                                                      \
      const unsigned vz = pv ? (unsigned)(zbase + prow * OHW) * 4u : kPoison;                        \
      /* rows past M (last, partial M tile) re-read the last valid 8-row group: their products land */ \
      /* in slab rows >= M that nobody reads -- no per-row branch or select in the loop            */ \
      _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                            \
        ra[i] = buf_f32(rz, vz, min(m0 + i * 8, g.M - 8) * OHW * 4);                                 \
      if (do_sums) {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) rsum[i] += ra[i];                        \
      }                                                                                              \
      const int ih = ih0 + e0.y, iw = iw0 + e0.z;                                                    \
      const bool ok = pv & ((unsigned)ih < (unsigned)g.H) & ((unsigned)iw < (unsigned)g.W);          \
      const unsigned vx = ok ? (unsigned)(xbase + (e0.x - e0.w) + prow * planeHW) * 4u : kPoison;    \
      _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) rb[i] = buf_f32(rx, vx, (e0.w + i * 8 * planeHW) * 4); \
    } else {                                                                                         \
      _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                          \
        const int m = m0 + prow + i * 8;                                                             \
        ra[i] = buf_f32(rz, (pv & (m < g.M)) ? (unsigned)(zbase + m * OHW) * 4u : kPoison, 0);       \
        if (do_sums) rsum[i] += ra[i];                                                               \
      }                                                                                              \
      _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) {                                          \
        const int ih = ih0 + (te_d[i] >> 16), iw = iw0 + (int)(short)(te_d[i] & 0xffff);             \
        const bool ok = pv & ((unsigned)ih < (unsigned)g.H) & ((unsigned)iw < (unsigned)g.W);        \
        rb[i] = buf_f32(rx, ok ? (unsigned)(xbase + te_off[i]) * 4u : kPoison, 0);                   \
      }                                                                                              \
    }                                                                                                \
    /* advance the cursor by one step */                                                             \
    pix += kWgPix;                                                                                   \
    pow_ += kWgPix;                                                                                  \
    while (pow_ >= g.OW) { pow_ -= g.OW; ++poh; }                                                    \
    while (poh >= g.OH) { poh -= g.OH; ++pn; }                                                       \
  }
#define DASAC_WG_STORE(buf)                                                                          \
  if (X3) {                                                                                          \
    unsigned short* a16 = reinterpret_cast<unsigned short*>(&sA[buf][0]);                            \
    unsigned short* b16 = reinterpret_cast<unsigned short*>(&sB[buf][0]);                            \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                            \
      const unsigned h = pack_bf16(0.f, ra[i]);                                                      \
      const unsigned l = pack_bf16(0.f, ra[i] - __uint_as_float(h));                                 \
      const int e = (((pl >> 3) * PA + prow + i * 8) << 3) + (pl & 7);                               \
      a16[e] = (unsigned short)(h >> 16);                                                            \
      a16[32 * PA + e] = (unsigned short)(l >> 16);                                                  \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) {                                            \
      const unsigned h = pack_bf16(0.f, rb[i]);                                                      \
      const unsigned l = pack_bf16(0.f, rb[i] - __uint_as_float(h));                                 \
      const int e = (((pl >> 3) * PB + prow + i * 8) << 3) + (pl & 7);                               \
      b16[e] = (unsigned short)(h >> 16);                                                            \
      b16[32 * PB + e] = (unsigned short)(l >> 16);                                                  \
    }                                                                                                \
  } else {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) sA[buf][(prow + i * 8) * kWgPitch + pl] = ra[i]; \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) sB[buf][(prow + i * 8) * kWgPitch + pl] = rb[i]; \
  }

  if (steps > 0) {
    DASAC_WG_LOAD();
    DASAC_WG_STORE(0);
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) DASAC_WG_LOAD();
    if (X3) {
      const f32x4* a4 = reinterpret_cast<const f32x4*>(&sA[0][0]) + wm * WM + li;
      const f32x4* b4 = reinterpret_cast<const f32x4*>(&sB[0][0]) + wn * WN + li;
#pragma unroll
      for (int kb = 0; kb < kWgPix / 16; ++kb) {
        const int oct = 2 * kb + lh;                     // this lane half's 8 pixels of the 16-pixel MFMA block
        f32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = a4[oct * PA + i * 32];
          al[i] = a4[(4 + oct) * PA + i * 32];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = b4[oct * PB + j * 32];
          bl[j] = b4[(4 + oct) * PB + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(al[i]), as_bf16x8(bh[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ah[i]), as_bf16x8(bl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ah[i]), as_bf16x8(bh[j]), acc[i][j], 0, 0, 0);
          }
      }
    } else {
    // one ds_read_b128 per operand feeds FOUR MFMAs: lane half lh holds pixels 8g + 4*lh + {0..3} of its row and MFMA e
    // contracts the pixel pair {8g + e, 8g + 4 + e} (any pairing is valid as long as both operands agree)
    const float* a_base = &sA[0][(wm * WM + li) * kWgPitch + 4 * lh];
    const float* b_base = &sB[0][(wn * WN + li) * kWgPitch + 4 * lh];
#pragma unroll
    for (int gq = 0; gq < kWgPix / 8; ++gq) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * kWgPitch + 8 * gq);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * kWgPitch + 8 * gq);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    }
    __syncthreads();
    if (s + 1 < steps) {
      DASAC_WG_STORE(0);
      __syncthreads();
    }
  }
#undef DASAC_WG_LOAD
#undef DASAC_WG_STORE

  if (do_sums) {
    // the 32 lanes of a half-wave hold the 32 pixels of the same rows: butterfly inside the half
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      float v = rsum[i];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (pl == 0) Psum[(size_t)split * g.Mpad + m0 + prow + i * 8] = v;
    }
  }
  }   // !QUAD
  // partial slab [split][Mpad][Kpad], k contiguous
  float* slab = P + (size_t)split * g.Mpad * g.Kpad;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = kb0 + wn * WN + j * 32 + li;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int m = m0 + wm * WM + i * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * lh;
        slab[(size_t)m * g.Kpad + k] = acc[i][j][rg];
      }
  }
}

// ------------------------------------------------------------------------------------------
// small helpers: gather table, weight packing, wgrad slab reduction
// ------------------------------------------------------------------------------------------
struct TapGroups {
  int n;             // number of branches (1, or 4 for the fused ASPP)
  int kh[4], kw[4], dil[4], pad[4];
};

// table row k -> (tap, c); tap enumerates (branch, kh, kw).
//   order 0 (tap-major):    k = tap*C + c                         -- weight-gradient tiles want one tap per 128 rows
//   order 1 (chunk-major):  k = ((c/16)*taps + tap)*16 + c%16     -- forward / data gradient: consecutive K-steps
//       walk the taps of the SAME 16 channel planes, so the shifted re-reads of a dilated 3x3 hit L2
//       instead of streaming the whole channel range between two taps (needs C % 16 == 0)
__device__ __forceinline__ void decode_k(int k, int C, int taps, int order, int& tap, int& c) {
  if (order == 0) {
    tap = k / C;
    c = k - tap * C;
  } else {
    const int blk = k >> 4, chunk = blk / taps;
    tap = blk - chunk * taps;
    c = chunk * 16 + (k & 15);
  }
}
__device__ __forceinline__ int encode_k(int tap, int c, int C, int taps, int order) {
  return order == 0 ? tap * C + c : (((c >> 4) * taps + tap) << 4) + (c & 15);
}

__global__ void build_table(int4* tab, TapGroups tg, int C, int K, int Kpad, int planeHW, int W, int sign, int taps,
                            int order) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  int4 e = make_int4(0, kInvalid, kInvalid, 0);
  if (k < K) {
    int tap, c;
    decode_k(k, C, taps, order, tap, c);
    int b = 0;
    while (b < tg.n - 1 && tap >= tg.kh[b] * tg.kw[b]) {
      tap -= tg.kh[b] * tg.kw[b];
      ++b;
    }
    const int kh = tap / tg.kw[b], kw = tap - kh * tg.kw[b];
    const int dh = sign * (kh * tg.dil[b] - tg.pad[b]), dw = sign * (kw * tg.dil[b] - tg.pad[b]);
    e = make_int4(c * planeHW + dh * W + dw, dh, dw, c * planeHW);
  }
  tab[k] = e;
}

// mode 0 (forward):  Wp[(tap0+tap)*Cin + ci][co] = W[co][ci][tap] * (scale ? scale[co] : 1)
// mode 1 (dgrad):    Wp[(tap0+tap)*Cout + co][ci] = W[co][ci][tap] * (scale ? scale[co] : 1)
__global__ void pack_weights(const float* __restrict__ Wt, const float* __restrict__ scale, float* __restrict__ Wp,
                             int Cout, int Cin, int taps, int tap0, int total_taps, int Mpad, int mode, int order) {
  const int C = mode == 0 ? Cin : Cout;
  const int rows = taps * C;
  const int64_t total = (int64_t)rows * Mpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / Mpad), m = (int)(i - (int64_t)row * Mpad);
    const int tap = row / C, c = row - tap * C;
    float v = 0.f;
    if (mode == 0) {
      if (m < Cout) {
        v = Wt[((int64_t)m * Cin + c) * taps + tap];
        if (scale) v = v * scale[m];
      }
    } else {
      if (m < Cin) {
        v = Wt[((int64_t)c * Cin + m) * taps + tap];
        if (scale) v = v * scale[c];
      }
    }
    const int64_t k = encode_k(tap0 + tap, c, C, total_taps, order);
    Wp[((k >> 2) * Mpad + m) * 4 + (k & 3)] = v;   // k-interleaved: [(k/4)][m][k%4]
  }
}

// All (single-branch) convolutions of a network in ONE launch: job j describes one packed operand (forward or data-gradient
// layout of one layer), chunk (j, c) covers kPackChunk consecutive floats of its output -- the K padding rows and the M padding
// columns are written as zeros here, so no memset per layer either.  Replaces ~2 x 104 pack_weights launches per optimiser step.
constexpr int kPackChunk = 256 * 32;
__global__ __launch_bounds__(256) void pack_weights_multi(const dasac_pack_job* __restrict__ jobs, const int2* __restrict__ chunks) {
  const int2 ch = chunks[blockIdx.x];
  const dasac_pack_job jb = jobs[ch.x];
  const int C = jb.mode == 0 ? jb.Cin : jb.Cout;
  const int K = jb.taps * C;
  const int64_t total = (int64_t)jb.Kpad * jb.Mpad;
  const int64_t base = (int64_t)ch.y * kPackChunk;
  const int row4 = jb.Mpad * 4;
#pragma unroll 4
  for (int u = 0; u < kPackChunk / 256; ++u) {
    const int64_t j = base + u * 256 + threadIdx.x;             // == offset in the packed operand [(k/4)][Mpad][4]
    if (j >= total) break;
    const int kq = (int)(j / row4), rem = (int)(j - (int64_t)kq * row4);
    const int m = rem >> 2, k = kq * 4 + (rem & 3);
    float v = 0.f;
    if (k < K) {
      int tap, c;
      decode_k(k, C, jb.taps, jb.order, tap, c);
      if (jb.mode == 0) {
        if (m < jb.Cout) {
          v = jb.w[((int64_t)m * jb.Cin + c) * jb.taps + tap];
          if (jb.scale) v = v * jb.scale[m];
        }
      } else if (m < jb.Cin) {
        v = jb.w[((int64_t)c * jb.Cin + m) * jb.taps + tap];
        if (jb.scale) v = v * jb.scale[c];
      }
    }
    jb.out[j] = v;
  }
}

// fp32 packed weights [(k/4)][Mpad][4] -> split-bf16 operands: K-step t (16 k) holds four slots of Mpad
// 16-byte words, slot 2*o + h = octet o (k = 16t + 8o .. +7), h = 0 heads / 1 tails -- the same addressing
// as the fp32 tile (slot == quad), so conv_gemm's weight path is shared.
__global__ void pack_x3(const f32x4* Wp, f32x4* Wx, int Mpad, int Kpad) {   // Wx may alias Wp (in-place conversion)
  const int64_t total = (int64_t)(Kpad / 8) * Mpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t oct = i / Mpad;                  // global octet index = 2*t + o
    const int m = (int)(i - oct * Mpad);
    f32x4 heads, tails;
    split_bf16(Wp[(2 * oct) * Mpad + m], Wp[(2 * oct + 1) * Mpad + m], heads, tails);
    Wx[(2 * oct) * Mpad + m] = heads;
    Wx[(2 * oct + 1) * Mpad + m] = tails;
  }
}

// sum over the pixel splits of one slab element: four independent chains so that the loads of several slabs are in
// flight together (the slabs are 1-64 separate HBM streams)
__device__ __forceinline__ float sum_splits(const float* __restrict__ p, size_t slab, int splits) {
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
  int s = 0;
  for (; s + 4 <= splits; s += 4) {
    g0 += p[(size_t)s * slab];
    g1 += p[(size_t)(s + 1) * slab];
    g2 += p[(size_t)(s + 2) * slab];
    g3 += p[(size_t)(s + 3) * slab];
  }
  for (; s < splits; ++s) g0 += p[(size_t)s * slab];
  return (g0 + g1) + (g2 + g3);
}

// dW[co][ci][tap] = scale[co] * sum_s P[s][co][(tap0+tap)*Cin+ci];  dot[row][co] = this block's part of sum W*G (unscaled G);
// sum_dz[co] = sum_s Psum[s][co]
// grid (chunks of 64 input channels, output channels), 256 threads = 64 channels x 4 split lanes.  A thread adds the splits
// s = lane, lane + 4, ... of up to 9 taps of ITS channel (9 independent, coalesced slab loads in flight per step); the four
// lanes meet in LDS and the block writes its 64 x taps gradients in dW's own order -- one contiguous run per block, where the
// tap-major slab order would touch every 36-byte [ci] group of a 3x3 layer nine times from nine places.  The dot term
// leaves as one PARTIAL per block, dot[blockIdx.x][co] (dasac_conv_wgrad_dot_rows rows): bn_param_grads adds the rows in a fixed
// order, so the gamma gradient is bit-identical from run to run (rounds 1-3 combined them with float atomics).
constexpr int kWrTaps = 9;
__global__ __launch_bounds__(256) void wgrad_reduce_scalar(const float* __restrict__ P, const float* __restrict__ Psum, int splits,
                                                    int Mpad, int Kpad, const float* __restrict__ Wt,
                                                    const float* __restrict__ scale, float* __restrict__ dW,
                                                    float* __restrict__ dot, float* __restrict__ sum_dz, int Cin, int taps,
                                                    int tap0, int flat_cin) {
  // flat_cin > 0 (few input channels: the stem, VGG's first conv): the caller passes Cin = true Cin * taps and taps = 1, the
  // 64 channel lanes then run along the slab's whole k axis (a 3-channel layer would use 3 of them); dW's index is remapped.
  const int co = blockIdx.y;
  const int tx = threadIdx.x, c = tx & 63, q = tx >> 6;
  if (sum_dz && blockIdx.x == 0 && q == 0) {                 // one wave: the splits' channel sums in parallel
    float sv = 0.f;
    for (int s2 = c; s2 < splits; s2 += 64) sv += Psum[(size_t)s2 * Mpad + co];
    sv = wave_sum(sv);
    if (c == 0) sum_dz[co] = sv;
  }
  __shared__ float red[4][kWrTaps][64];
  const int ci0 = blockIdx.x * 64, ci = ci0 + c;
  const float sc = scale ? scale[co] : 1.f;
  const size_t slab = (size_t)Mpad * Kpad;
  float part = 0.f;
  for (int tb = 0; tb < taps; tb += kWrTaps) {
    const int nt = min(kWrTaps, taps - tb);
    float acc[kWrTaps];
#pragma unroll
    for (int u = 0; u < kWrTaps; ++u) acc[u] = 0.f;
    if (ci < Cin) {
      const float* p0 = P + (size_t)co * Kpad + (size_t)(tap0 + tb) * Cin + ci;
#pragma unroll 4
      for (int s2 = q; s2 < splits; s2 += 4) {         // unrolled: the loads of four splits are issued before the first add
        const float* ps = p0 + (size_t)s2 * slab;
#pragma unroll
        for (int u = 0; u < kWrTaps; ++u)
          if (u < nt) acc[u] += ps[(size_t)u * Cin];
      }
    }
#pragma unroll
    for (int u = 0; u < kWrTaps; ++u) red[q][u][c] = acc[u];
    __syncthreads();
    for (int e = tx; e < 64 * nt; e += 256) {                // (channel, tap) pairs in dW order
      const int c2 = e / nt, u2 = e - c2 * nt;
      if (ci0 + c2 < Cin) {
        const float gsum = (red[0][u2][c2] + red[1][u2][c2]) + (red[2][u2][c2] + red[3][u2][c2]);
        size_t widx = ((size_t)co * Cin + ci0 + c2) * taps + tb + u2;
        if (flat_cin > 0) {                                    // k = tap * Cin_true + ci  ->  [co][ci][tap]
          const int k = ci0 + c2, tap = k / flat_cin, ci_t = k - tap * flat_cin;
          widx = ((size_t)co * flat_cin + ci_t) * (Cin / flat_cin) + tap;
        }
        if (dot) part += gsum * Wt[widx];
        dW[widx] = gsum * sc;
      }
    }
    __syncthreads();
  }
  if (dot) {
    __shared__ float redd[4];
    part = wave_sum(part);
    if (c == 0) redd[q] = part;
    __syncthreads();
    if (tx == 0) dot[(size_t)blockIdx.x * gridDim.y + co] = (redd[0] + redd[1]) + (redd[2] + redd[3]);   // partial row: no atomics
  }
}

// The same reduction for layers whose channel count is a multiple of 64 (every layer that matters): a thread owns FOUR consecutive
// input channels (one 16-byte slab load per tap and split) and the 256 threads are 16 channel quads x 16 split lanes -- four times
// the bytes in flight per thread of the scalar kernel above (which moved its slabs at 1.5 TB/s).  The four split lanes of a wave
// meet in a butterfly, the four waves in LDS; output and dot term exactly as above.
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ P, const float* __restrict__ Psum, int splits,
                                                    int Mpad, int Kpad, const float* __restrict__ Wt,
                                                    const float* __restrict__ scale, float* __restrict__ dW,
                                                    float* __restrict__ dot, float* __restrict__ sum_dz, int Cin, int taps,
                                                    int tap0) {
  const int co = blockIdx.y;
  const int tx = threadIdx.x, lane = tx & 63, wave = tx >> 6;
  const int c4 = lane & 15, q = wave * 4 + (lane >> 4);      // channel quad, split lane 0..15
  if (sum_dz && blockIdx.x == 0 && wave == 0) {              // one wave: the splits' channel sums in parallel
    float sv = 0.f;
    for (int s2 = lane; s2 < splits; s2 += 64) sv += Psum[(size_t)s2 * Mpad + co];
    sv = wave_sum(sv);
    if (lane == 0) sum_dz[co] = sv;
  }
  __shared__ float red[4][kWrTaps][64];
  const int ci0 = blockIdx.x * 64;
  const float sc = scale ? scale[co] : 1.f;
  const size_t slab = (size_t)Mpad * Kpad;
  float part = 0.f;
  for (int tb = 0; tb < taps; tb += kWrTaps) {
    const int nt = min(kWrTaps, taps - tb);
    f32x4 acc[kWrTaps];
#pragma unroll
    for (int u = 0; u < kWrTaps; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p0 = P + (size_t)co * Kpad + (size_t)(tap0 + tb) * Cin + ci0 + 4 * c4;      // 16-byte aligned: Kpad, Cin, ci0 % 64 == 0
#pragma unroll 2
    for (int s2 = q; s2 < splits; s2 += 16) {
      const float* ps = p0 + (size_t)s2 * slab;
#pragma unroll
      for (int u = 0; u < kWrTaps; ++u)
        if (u < nt) acc[u] += *reinterpret_cast<const f32x4*>(ps + (size_t)u * Cin);
    }
#pragma unroll
    for (int u = 0; u < kWrTaps; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[u][e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[u][e] = v;
      }
      if (lane < 16) *reinterpret_cast<f32x4*>(&red[wave][u][4 * c4]) = acc[u];
    }
    __syncthreads();
    for (int e = tx; e < 64 * nt; e += 256) {                // (channel, tap) pairs in dW order
      const int c2 = e / nt, u2 = e - c2 * nt;
      const float gsum = (red[0][u2][c2] + red[1][u2][c2]) + (red[2][u2][c2] + red[3][u2][c2]);
      const size_t widx = ((size_t)co * Cin + ci0 + c2) * taps + tb + u2;
      if (dot) part += gsum * Wt[widx];
      dW[widx] = gsum * sc;
    }
    __syncthreads();
  }
  if (dot) {
    __shared__ float redd[4];
    part = wave_sum(part);
    if (lane == 0) redd[wave] = part;
    __syncthreads();
    if (tx == 0) dot[(size_t)blockIdx.x * gridDim.y + co] = (redd[0] + redd[1]) + (redd[2] + redd[3]);   // partial row: no atomics
  }
}

// The same reduction for ONE-tap layers (every 1x1 convolution: half of the step's weight-gradient launches).  With taps == 1 the slab
// order IS dW's order, so no transposition through LDS is needed, and the 16 x 16 arrangement above leaves a thread three 16-byte loads
// at 48 splits (measured stand-alone: 25 us for 50 MB = 2.0 TB/s, the 2048 x 512 layer 68 us = 0.74 TB/s over 16 384 blocks).  Here a block
// owns one output channel and 256 input channels -- 64 channel quads x 4 split lanes, a wave per split lane --, a thread streams
// splits / 4 slabs with four loads in flight; the four waves meet in LDS and wave w then holds 64-channel chunk w of the block: its dot
// partial comes out of the same wave butterfly as the general kernel's, row (4 * blockIdx.x + w).
__global__ __launch_bounds__(256) void wgrad_reduce_1x1(const float* __restrict__ P, const float* __restrict__ Psum, int splits, int Mpad,
                                                        int Kpad, const float* __restrict__ Wt, const float* __restrict__ scale,
                                                        float* __restrict__ dW, float* __restrict__ dot, float* __restrict__ sum_dz,
                                                        int Cin) {
  const int co = blockIdx.y;
  const int tx = threadIdx.x, lane = tx & 63, wave = tx >> 6;
  if (sum_dz && blockIdx.x == 0 && wave == 0) {              // one wave: the splits' channel sums in parallel
    float sv = 0.f;
    for (int s2 = lane; s2 < splits; s2 += 64) sv += Psum[(size_t)s2 * Mpad + co];
    sv = wave_sum(sv);
    if (lane == 0) sum_dz[co] = sv;
  }
  __shared__ __attribute__((aligned(16))) float red[4][256];
  const int ci0 = blockIdx.x * 256, cq = ci0 + 4 * lane;     // this thread's channel quad
  const size_t slab = (size_t)Mpad * Kpad;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (cq < Cin) {
    const float* p0 = P + (size_t)co * Kpad + cq;
#pragma unroll 4
    for (int s2 = wave; s2 < splits; s2 += 4) acc += *reinterpret_cast<const f32x4*>(p0 + (size_t)s2 * slab);
  }
  *reinterpret_cast<f32x4*>(&red[wave][4 * lane]) = acc;
  __syncthreads();
  const int ci = ci0 + tx;                                   // one channel per thread now; wave w = 64-channel chunk w of the block
  float part = 0.f;
  if (ci < Cin) {
    const float gsum = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    const size_t widx = (size_t)co * Cin + ci;
    if (dot) part = gsum * Wt[widx];
    dW[widx] = gsum * (scale ? scale[co] : 1.f);
  }
  if (dot && ci0 + 64 * wave < Cin) {
    part = wave_sum(part);
    if (lane == 0) dot[(size_t)(4 * blockIdx.x + wave) * gridDim.y + co] = part;
  }
}

static int fill_geom(GemmGeom& g, int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int Mpad, int Kpad,
                     int OutH, int OutW, int ostride) {
  DASAC_REQUIRE(Nb > 0 && Cx > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && stride > 0 && M > 0, "conv: bad geometry");
  DASAC_REQUIRE((int64_t)Nb * Cx * H * W < (1ll << 30) && (int64_t)Nb * M * OutH * OutW < (1ll << 30) &&
                    (int64_t)Nb * OH * OW < (1ll << 30),
                "conv: tensor exceeds 2^30 elements (32-bit byte offsets)");
  DASAC_REQUIRE(Kpad % 16 == 0 && Mpad % 32 == 0 && Mpad >= M, "conv: bad padding Kpad=%d Mpad=%d", Kpad, Mpad);
  g.H = H; g.W = W; g.CxHW = Cx * H * W; g.OH = OH; g.OW = OW; g.stride = stride; g.Npix = Nb * OH * OW; g.n_tile0 = 0;
  g.M = M; g.Mpad = Mpad; g.Kpad = Kpad; g.OutH = OutH; g.OutW = OutW; g.ostride = ostride;
  DASAC_REQUIRE((int64_t)Nb * Cx * H * W * 4 <= kMaxTensorBytes && (int64_t)Nb * M * OH * OW * 4 <= kMaxTensorBytes,
                "conv: tensor exceeds the 4 GiB buffer-descriptor window");
  DASAC_REQUIRE((int64_t)Nb * M * OutH * OutW * 4 <= kMaxTensorBytes, "conv: output exceeds the 4 GiB buffer-descriptor window");
  // (row / plane offsets inside ONE image travel as signed scalar offsets)
  DASAC_REQUIRE((int64_t)Cx * H * W * 4 < (1ll << 31) && (int64_t)M * OutH * OutW * 4 < (1ll << 31) && (int64_t)M * OH * OW * 4 < (1ll << 31),
                "conv: one image of a tensor exceeds 2 GiB");
  g.x_bytes = (unsigned)((int64_t)Nb * Cx * H * W * 4);
  g.z_bytes = (unsigned)((int64_t)Nb * M * OH * OW * 4);
  g.out_bytes = (unsigned)((int64_t)Nb * M * OutH * OutW * 4);
  g.w_bytes = 0;
  return DASAC_OK;
}


// stream-K pays when the tile count leaves the last round of resident blocks mostly empty.
// The persistent grid (3 workers per CU x 256 CUs) and the XCD mapping are sized for the full MI355X; a device with
// fewer CUs (partitioned modes, other parts) gets the tile-per-block schedule only.  The answer is per device id
// (a process may drive several devices / switch with hipSetDevice).
static bool persistent_grid_fits() {
  constexpr int kMaxDev = 64;
  static std::atomic<signed char> cache[kMaxDev];          // 0 unknown, 1 fits, -1 does not
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const bool cached = dev >= 0 && dev < kMaxDev;
  if (cached && cache[dev].load(std::memory_order_relaxed) != 0) return cache[dev].load(std::memory_order_relaxed) > 0;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  const bool ok = cus >= kNumCu;
  if (cached) cache[dev].store(ok ? 1 : -1, std::memory_order_relaxed);
  return ok;
}

// workers of a persistent stream-K launch: 3 per CU on the CUs this process does not leave to overlapped collectives.  `reserved` is
// read ONCE per launch decision (reserved_cus()) and handed through: a concurrent dasac_set_reserved_cus must not make the
// eligibility test, the schedule choice and the grid size of one launch disagree.
static int sk_workers(int reserved) { return (kNumCu - reserved) * kSkWorkersPerCu; }

static bool want_streamk(int tiles, int k_steps, int reserved) {
  if (!persistent_grid_fits()) return false;
  if ((long long)tiles * k_steps < sk_workers(reserved)) return false;                  // every range gets >= 1 K-step
  // measured: the persistent schedule (3 workers/CU, <=168 registers) wins on long contractions whose tile count
  // fills the last round of the plain launch badly; short-K 1x1 layers are better off with the plain kernel's
  // 4 blocks per CU (128 registers) even with a partly empty last round (85 vs 99 TFLOP/s at K = 256) -- unless
  // the launch would leave most of the chip idle (small batches: 296 tiles at B = 2, 148 at B = 1).
  const int resident = (kNumCu - reserved) * 3;
  const int rounds = (tiles + resident - 1) / resident;
  const double eff = (double)tiles / ((double)rounds * resident);
  return k_steps >= 64 ? eff < 0.93 : (k_steps >= 16 && eff < 0.6);
}

// Split-K tail (SK == 2) of a launch over ALL n_tiles pixel tiles: the leading whole rounds of resident blocks run one block per
// tile, the remaining R tiles are cut into `split` K-ranges so that the pieces fill (a whole number of) rounds again.  Chosen for
// the shapes the persistent schedule was chosen for in rounds 2-5 (want_streamk) when at least one whole round leads.  The split
// minimises rounds x (K-steps per piece + a piece's fixed cost: prologue, deposit or epilogue ~ 6 K-steps).  Returns 0 = no tail.
static int tail_plan(int m_tiles, int n_tiles, int k_steps, int reserved, int& lead_n) {
  const int tiles = m_tiles * n_tiles;
  if (!want_streamk(tiles, k_steps, reserved)) return 0;
  lead_n = (tiles / kTileSlots) * kTileSlots / m_tiles;          // pixel tiles of the leading whole rounds
  // (fewer tiles than resident blocks -- small batches, cfg-2's 296 tiles -- stay on the persistent kernel: the same launches as pure
  // split-K pieces of this kernel, 4 blocks per CU, measured 35.4 against 33.9 ms per cfg-2 step: equal ranges balance better than
  // equal pieces when nothing leads; profiles/EXPERIMENTS.md round 6)
  if (lead_n <= 0 || lead_n >= n_tiles) return 0;
  const int R = (n_tiles - lead_n) * m_tiles;
  int best = 0;
  double best_cost = 0.0;
  // (a sweep of the minimum piece length 8 .. 32 K-steps and of the fixed cost 6 .. 20 moved the three tail shapes of the step by
  // less than 1 %: tools/experiments/r6_tail_sweep.sh, EXPERIMENTS.md)
  constexpr int min_steps = 8;
  constexpr double ovh = 6.0;
  for (int sp = 1; sp <= 8 && sp * min_steps <= k_steps; ++sp) {
    if ((long long)R * (sp - 1) > kTailSlots) break;
    const int rounds = (R * sp + kTileSlots - 1) / kTileSlots;
    const double cost = rounds * ((double)k_steps / sp + ovh);
    if (best == 0 || cost < best_cost - 1e-9) {
      best = sp;
      best_cost = cost;
    }
  }
  return best;
}

// `n_tiles` pixel tiles starting at g.n_tile0; schedule 0 = pick (split-K tail over a whole tensor, else want_streamk), 1 = one
// block per tile, 2 = stream-K
template <int BM, int BN, int WAVES_M, int BK, bool FAST, bool X3 = false, int BITS = 0>
static int launch_gemm(const float* X, const float* Wp, const int4* tab, float* Out, const GemmGeom& g,
                       const Epilogue& ep, int n_tiles, int schedule, void* workspace, size_t ws_bytes, hipStream_t s) {
  const int m_tiles = (g.M + BM - 1) / BM;
  if ((long long)m_tiles * (n_tiles + kNumXcd) * (g.Kpad / BK) >= (1ll << 31)) return fail(DASAC_EINVAL, "conv_gemm: iteration space exceeds 2^31");
  const int reserved = reserved_cus();
  const size_t part_bytes = (size_t)(kTailSlots > kSkWorkers ? kTailSlots : kSkWorkers) * (BM * BN) * sizeof(float);
  const size_t need = part_bytes + (size_t)(kTailSlots + 1) * sizeof(int);
  float* partial = reinterpret_cast<float*>(workspace);
  int* flags = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + part_bytes);
  if constexpr (BM == 128) {
    const bool whole = g.n_tile0 == 0 && (long long)n_tiles * BN >= g.Npix;
    int lead_n = 0;
    const int split = (schedule == 0 && whole && workspace && persistent_grid_fits()) ? tail_plan(m_tiles, n_tiles, g.Kpad / BK, reserved, lead_n) : 0;
    if (split > 0) {
      if (ws_bytes < need) return fail(DASAC_EWORKSPACE, "conv_gemm: workspace too small (%zu < %zu)", ws_bytes, need);
      const int R = (n_tiles - lead_n) * m_tiles;
      const int lead_pad = (lead_n + kNumXcd - 1) / kNumXcd * kNumXcd, r_pad = (R + kNumXcd - 1) / kNumXcd * kNumXcd;
      hipLaunchKernelGGL((conv_gemm<BM, BN, WAVES_M, BK, FAST, 2, X3, BITS>), dim3(lead_pad * m_tiles + r_pad * split), dim3(kThreads), 0, s, X, Wp,
                         tab, Out, g, ep, m_tiles, lead_n, partial, flags, R, split);
      return DASAC_OK;
    }
  }
  const bool sk_ok = BM == 128 && workspace && (long long)m_tiles * n_tiles * (g.Kpad / BK) >= sk_workers(reserved) && persistent_grid_fits();
  if (sk_ok && (schedule == 2 || (schedule == 0 && want_streamk(m_tiles * n_tiles, g.Kpad / BK, reserved)))) {
    if (ws_bytes < need) return fail(DASAC_EWORKSPACE, "conv_gemm: workspace too small (%zu < %zu)", ws_bytes, need);
    hipLaunchKernelGGL((conv_gemm<BM, BN, WAVES_M, BK, FAST, 1, X3, BITS>), dim3(sk_workers(reserved)), dim3(kThreads), 0, s, X, Wp, tab, Out, g,
                       ep, m_tiles, n_tiles, partial, flags, 0, 1);
    return DASAC_OK;
  }
  const int n_tiles_pad = (n_tiles + kNumXcd - 1) / kNumXcd * kNumXcd;
  hipLaunchKernelGGL((conv_gemm<BM, BN, WAVES_M, BK, FAST, 0, X3, BITS>), dim3(n_tiles_pad * m_tiles), dim3(kThreads), 0, s, X, Wp,
                     tab, Out, g, ep, m_tiles, n_tiles, nullptr, nullptr, 0, 1);
  return DASAC_OK;
}

template <int BM, int BN, int WAVES_M, bool FAST>
static void launch_wgrad(bool x3, const float* dZ, const float* X, const int4* tab, float* P, float* Psum, const GemmGeom& g,
                         int splits, int pix_per_split, hipStream_t s) {
  const int m_tiles = g.Mpad / BM, k_tiles = g.Kpad / BN;
  const int grid = (m_tiles * k_tiles * splits + kNumXcd - 1) / kNumXcd * kNumXcd;
  if (x3)
    hipLaunchKernelGGL((conv_wgrad<BM, BN, WAVES_M, FAST, true>), dim3(grid), dim3(kThreads), 0, s, dZ, X, tab, P,
                       Psum, g, m_tiles, k_tiles, splits, pix_per_split);
  else
    hipLaunchKernelGGL((conv_wgrad<BM, BN, WAVES_M, FAST, false>), dim3(grid), dim3(kThreads), 0, s, dZ, X, tab, P,
                       Psum, g, m_tiles, k_tiles, splits, pix_per_split);
}

}  // namespace dasac

using namespace dasac;

// M tile chosen from the padded channel count: 128 | 64 | 32
static int pick_bm(int Mpad) { return Mpad % 128 == 0 ? 128 : (Mpad % 64 == 0 ? 64 : 32); }

extern "C" int dasac_conv_mpad(int M) { return M > 64 ? (M + 127) / 128 * 128 : (M > 32 ? 64 : 32); }
extern "C" int dasac_conv_kpad(int K) { return (K + 127) / 128 * 128; }

extern "C" int dasac_conv_table(const int32_t* kh, const int32_t* kw, const int32_t* dil, const int32_t* pad,
                                int n_branches, int C, int plane_h, int plane_w, int transposed, int order, int32_t* table,
                                dasac_stream_t stream) {
  DASAC_REQUIRE(kh && kw && dil && pad && table, "conv_table: null pointer");
  DASAC_REQUIRE(n_branches >= 1 && n_branches <= 4 && C > 0, "conv_table: bad branches/C");
  TapGroups tg;
  tg.n = n_branches;
  int taps = 0;
  for (int b = 0; b < n_branches; ++b) {
    tg.kh[b] = kh[b]; tg.kw[b] = kw[b]; tg.dil[b] = dil[b]; tg.pad[b] = pad[b];
    taps += kh[b] * kw[b];
  }
  for (int b = n_branches; b < 4; ++b) tg.kh[b] = tg.kw[b] = tg.dil[b] = tg.pad[b] = 1;
  const int K = taps * C, Kpad = dasac_conv_kpad(K);
  DASAC_REQUIRE(order == 0 || (order == 1 && C % 16 == 0), "conv_table: chunk-major order needs C %% 16 == 0");
  hipLaunchKernelGGL(build_table, dim3((Kpad + 255) / 256), dim3(256), 0, as_stream(stream), reinterpret_cast<int4*>(table),
                     tg, C, K, Kpad, plane_h * plane_w, plane_w, transposed ? -1 : 1, taps, order);
  DASAC_CHECK_LAUNCH("build_table");
  return DASAC_OK;
}

extern "C" int dasac_conv_pack(const float* w, const float* scale, int Cout, int Cin, int taps, int tap0, int total_taps,
                               int transposed, int order, float* packed, dasac_stream_t stream) {
  DASAC_REQUIRE(w && packed, "conv_pack: null pointer");
  const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
  const int Mpad = dasac_conv_mpad(M), K = total_taps * C, Kpad = dasac_conv_kpad(K);
  hipStream_t s = as_stream(stream);
  if (tap0 == 0 && Kpad > K) DASAC_HIP(hipMemsetAsync(packed, 0, (size_t)Kpad * Mpad * sizeof(float), s));   // zero K padding
  const int64_t total = (int64_t)taps * C * Mpad;
  DASAC_REQUIRE(order == 0 || (order == 1 && C % 16 == 0), "conv_pack: chunk-major order needs C %% 16 == 0");
  hipLaunchKernelGGL(pack_weights, dim3(stream_grid(total, 256)), dim3(256), 0, s, w, scale, packed, Cout, Cin, taps, tap0,
                     total_taps, Mpad, transposed ? 1 : 0, order);
  DASAC_CHECK_LAUNCH("pack_weights");
  return DASAC_OK;
}

extern "C" int dasac_pack_chunk_elems(void) { return kPackChunk; }

extern "C" int dasac_conv_pack_multi(const dasac_pack_job* jobs, const int32_t* chunks, int n_chunks, dasac_stream_t stream) {
  DASAC_REQUIRE(jobs && chunks && n_chunks > 0, "conv_pack_multi: bad arguments");
  hipLaunchKernelGGL(pack_weights_multi, dim3(n_chunks), dim3(256), 0, as_stream(stream), jobs, reinterpret_cast<const int2*>(chunks));
  DASAC_CHECK_LAUNCH("pack_weights_multi");
  return DASAC_OK;
}

// 1 when dasac_conv_gemm (given a workspace) runs this shape on the persistent stream-K schedule
extern "C" int dasac_conv_gemm_schedule(int Nb, int OH, int OW, int M, int K) {
  const int Mpad = dasac_conv_mpad(M);
  if (pick_bm(Mpad) != 128) return 0;
  const int tiles = ((M + 127) / 128) * ((Nb * OH * OW + 127) / 128);
  return want_streamk(tiles, (K + kBK - 1) / kBK, reserved_cus()) ? 1 : 0;
}

// (Rounds 2-5 issued a long-K conv whose tile count leaves a ragged last round as TWO launches -- whole rounds one block per tile,
// the remainder on the persistent stream-K schedule (dasac_conv_gemm_plan) -- so that the lockstep K walk of the leading rounds
// keeps sharing halo rows in L2.  Round 6 folded the remainder into the first launch: tail_plan / conv_gemm<SK = 2> above.
// Measured on one box, cfg-3 fused step: GEMM kernel time 273.5 -> 272.6 ms, 169 launches per step gone; a piece's prologue and
// deposit / epilogue cost what the second launch's ramp and hand-off did: profiles/r6_tail_and_qtap_ab.txt.)
extern "C" size_t dasac_conv_gemm_workspace(void) {
  return (size_t)(kTailSlots > kSkWorkers ? kTailSlots : kSkWorkers) * 128 * 128 * sizeof(float) + (size_t)(kTailSlots + 1) * sizeof(int);
}

// K-ranges per tail tile when dasac_conv_gemm (schedule 0, whole tensor, workspace given) runs this shape as ONE launch of whole
// rounds of tile-per-block workgroups + a split-K tail; 0 = it does not (then dasac_conv_gemm_schedule tells the rest)
extern "C" int dasac_conv_gemm_tail_split(int Nb, int OH, int OW, int M, int K) {
  const int Mpad = dasac_conv_mpad(M);
  if (pick_bm(Mpad) != 128 || !persistent_grid_fits()) return 0;
  int lead_n = 0;
  return tail_plan((M + 127) / 128, (int)(((int64_t)Nb * OH * OW + 127) / 128), (K + kBK - 1) / kBK, reserved_cus(), lead_n);
}

#ifdef DASAC_TRACE_TILES
extern "C" int dasac_debug_set_tile_trace(void* buffer) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(dasac::g_tile_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t dasac_relu_bits_words(int M, int64_t Npix) { return (size_t)M * (size_t)((Npix + 31) / 32); }
// 1 when dasac_conv_gemm (fp32) can record / consume bit masks for an output of M channels over a gathered tensor of Cx channels
extern "C" int dasac_conv_gemm_bits_ok(int M, int Cx) { return (pick_bm(dasac_conv_mpad(M)) == 128 && Cx % kBK == 0) ? 1 : 0; }

static int conv_gemm_impl(bool x3, const float* x, const float* packed, const int32_t* table, float* out, int Nb, int Cx, int H,
                          int W, int OH, int OW, int stride, int M, int K, int OutH, int OutW, int ostride, const float* shift,
                          const float* res, const float* mask, const uint32_t* mask_bits, uint32_t* relu_bits_out, int relu,
                          int pix_begin, int pix_count, int schedule, void* workspace, size_t ws_bytes, dasac_stream_t stream,
                          float* stats = nullptr) {
  DASAC_REQUIRE(x && packed && table && out, "conv_gemm: null pointer");
  DASAC_REQUIRE(!stats || (!x3 && !mask_bits && !relu_bits_out && dasac_conv_gemm_bits_ok(M, Cx) && ostride == 1),
                "conv_gemm_stats: needs the 128-row fp32 tile with Cx %% 16 == 0 (dasac_conv_gemm_stats_ok), no bit masks");
  DASAC_REQUIRE(!(mask && mask_bits), "conv_gemm: give the ReLU pattern as fp32 mask OR as bit mask");
  DASAC_REQUIRE(!(mask_bits || relu_bits_out) || (ostride == 1 && OutH == OH && OutW == OW),
                "conv_gemm: bit masks index the GEMM's own pixel axis (ostride must be 1)");
  DASAC_REQUIRE(!relu_bits_out || relu, "conv_gemm: relu_bits_out records the pattern of a ReLU epilogue");
  DASAC_REQUIRE(!(x3 && (mask_bits || relu_bits_out)), "conv_gemm_x3: bit masks are implemented for the fp32 kernel only");
  DASAC_REQUIRE(!stats || (!res && !relu), "conv_gemm_stats: the statistics epilogue stores the raw convolution (+ shift): no residual, no ReLU");
  GemmGeom g;
  const int Mpad = dasac_conv_mpad(M), Kloop = (K + kBK - 1) / kBK * kBK;   // table/pack are padded to 128 >= Kloop
  int rc = fill_geom(g, Nb, Cx, H, W, OH, OW, stride, M, Mpad, Kloop, OutH, OutW, ostride);
  if (rc) return rc;
  DASAC_REQUIRE((int64_t)dasac_conv_kpad(K) * Mpad * 4 < (1ll << 31), "conv: packed weights exceed 2 GiB");
  g.w_bytes = dasac_conv_kpad(K) * Mpad * 4;
  Epilogue ep{shift, res, mask, relu, mask_bits, relu_bits_out, (g.Npix + 31) / 32, stats};
  DASAC_REQUIRE((int64_t)M * ep.w32 * 4 < (1ll << 31), "conv_gemm: bit mask exceeds the 2 GiB buffer-descriptor window");
  const int4* tab = reinterpret_cast<const int4*>(table);
  hipStream_t s = as_stream(stream);
  const bool fast = Cx % kBK == 0;      // a K-step never straddles two taps
  const int bm = pick_bm(Mpad);
  // pixel range of this launch (whole tiles; the last one may be ragged only at the end of the tensor)
  const int bn = bm == 32 ? 256 : 128;
  const int pix_end = pix_count > 0 ? pix_begin + pix_count : g.Npix;
  DASAC_REQUIRE(schedule >= 0 && schedule <= 2, "conv_gemm: schedule must be 0 (auto), 1 (tile per block) or 2 (stream-K)");
  DASAC_REQUIRE(pix_begin >= 0 && pix_begin < g.Npix && pix_begin % bn == 0 && pix_end > pix_begin && pix_end <= g.Npix &&
                    (pix_end == g.Npix || pix_end % bn == 0),
                "conv_gemm: pixel range [%d, %d) must consist of whole %d-pixel tiles", pix_begin, pix_end, bn);
  g.n_tile0 = pix_begin / bn;
  const int n_tiles = (pix_end - pix_begin + bn - 1) / bn;
  if (x3) {
    DASAC_REQUIRE(bm >= 64, "conv_gemm_x3: needs M > 32 (use dasac_conv_gemm for skinny outputs)");
    if (bm == 128)
      rc = fast ? launch_gemm<128, 128, 2, kBK, true, true>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s)
                : launch_gemm<128, 128, 2, kBK, false, true>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s);
    else
      rc = fast ? launch_gemm<64, 128, 2, kBK, true, true>(x, packed, tab, out, g, ep, n_tiles, 1, nullptr, 0, s)
                : launch_gemm<64, 128, 2, kBK, false, true>(x, packed, tab, out, g, ep, n_tiles, 1, nullptr, 0, s);
    if (rc) return rc;
    DASAC_CHECK_LAUNCH("conv_gemm_x3");
    return DASAC_OK;
  }
  if (stats) {
    rc = launch_gemm<128, 128, 2, kBK, true, false, 3>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s);
    if (rc) return rc;
    DASAC_CHECK_LAUNCH("conv_gemm_stats");
    return DASAC_OK;
  }
  if (mask_bits || relu_bits_out) {
    DASAC_REQUIRE(dasac_conv_gemm_bits_ok(M, Cx) && !(mask_bits && relu_bits_out),
                  "conv_gemm: bit masks need the 128-row fp32 tile with Cx %% 16 == 0 (dasac_conv_gemm_bits_ok), one direction per call");
    rc = relu_bits_out ? launch_gemm<128, 128, 2, kBK, true, false, 1>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s)
                       : launch_gemm<128, 128, 2, kBK, true, false, 2>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s);
    if (rc) return rc;
    DASAC_CHECK_LAUNCH("conv_gemm");
    return DASAC_OK;
  }
  switch (bm) {
    case 128:
      rc = fast ? launch_gemm<128, 128, 2, kBK, true>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s)
                : launch_gemm<128, 128, 2, kBK, false>(x, packed, tab, out, g, ep, n_tiles, schedule, workspace, ws_bytes, s);
      break;
    case 64:
      rc = fast ? launch_gemm<64, 128, 2, kBK, true>(x, packed, tab, out, g, ep, n_tiles, 1, nullptr, 0, s)
                : launch_gemm<64, 128, 2, kBK, false>(x, packed, tab, out, g, ep, n_tiles, 1, nullptr, 0, s);
      break;
    default:
      rc = fast ? launch_gemm<32, 256, 1, kBK, true>(x, packed, tab, out, g, ep, n_tiles, 1, nullptr, 0, s)
                : launch_gemm<32, 256, 1, kBK, false>(x, packed, tab, out, g, ep, n_tiles, 1, nullptr, 0, s);
      break;
  }
  if (rc) return rc;
  DASAC_CHECK_LAUNCH("conv_gemm");
  return DASAC_OK;
}

extern "C" int dasac_conv_gemm(const float* x, const float* packed, const int32_t* table, float* out, int Nb, int Cx,
                               int H, int W, int OH, int OW, int stride, int M, int K, int OutH, int OutW, int ostride,
                               const float* shift, const float* res, const float* mask, const uint32_t* mask_bits,
                               uint32_t* relu_bits_out, int relu, int pix_begin, int pix_count, int schedule,
                               void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  return conv_gemm_impl(false, x, packed, table, out, Nb, Cx, H, W, OH, OW, stride, M, K, OutH, OutW, ostride, shift, res, mask,
                        mask_bits, relu_bits_out, relu, pix_begin, pix_count, schedule, workspace, ws_bytes, stream);
}

// dasac_conv_gemm whose epilogue also leaves per-tile channel statistics (batch-statistics BatchNorm behind the conv):
// stats [dasac_conv_gemm_stats_tiles(Nb, OH, OW)][2][dasac_conv_mpad(M)] floats = per 128-pixel tile the sum and the sum of
// squares of every output row as stored (shift / bias included) -- dasac_bn_train_finalize adds the tiles in a fixed order.
extern "C" int dasac_conv_gemm_stats_ok(int M, int Cx) { return dasac_conv_gemm_bits_ok(M, Cx); }
extern "C" int dasac_conv_gemm_stats_tiles(int Nb, int OH, int OW) { return (int)(((int64_t)Nb * OH * OW + 127) / 128); }
extern "C" int dasac_conv_gemm_stats(const float* x, const float* packed, const int32_t* table, float* out, int Nb, int Cx,
                                     int H, int W, int OH, int OW, int stride, int M, int K, int OutH, int OutW, int ostride,
                                     const float* shift, const float* res, int relu, int pix_begin, int pix_count, int schedule,
                                     void* workspace, size_t ws_bytes, float* stats, dasac_stream_t stream) {
  DASAC_REQUIRE(stats, "conv_gemm_stats: null statistics buffer");
  return conv_gemm_impl(false, x, packed, table, out, Nb, Cx, H, W, OH, OW, stride, M, K, OutH, OutW, ostride, shift, res, nullptr,
                        nullptr, nullptr, relu, pix_begin, pix_count, schedule, workspace, ws_bytes, stream, stats);
}

extern "C" int dasac_conv_gemm_x3(const float* x, const void* packed_x3, const int32_t* table, float* out, int Nb, int Cx,
                                  int H, int W, int OH, int OW, int stride, int M, int K, int OutH, int OutW, int ostride,
                                  const float* shift, const float* res, const float* mask, const uint32_t* mask_bits,
                                  uint32_t* relu_bits_out, int relu, int pix_begin, int pix_count, int schedule,
                                  void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  return conv_gemm_impl(true, x, reinterpret_cast<const float*>(packed_x3), table, out, Nb, Cx, H, W, OH, OW, stride, M, K, OutH,
                        OutW, ostride, shift, res, mask, mask_bits, relu_bits_out, relu, pix_begin, pix_count, schedule, workspace,
                        ws_bytes, stream);
}

extern "C" int dasac_conv_pack_x3(const float* packed, int M, int K, void* packed_x3, dasac_stream_t stream) {
  DASAC_REQUIRE(packed && packed_x3, "conv_pack_x3: null pointer");
  const int Mpad = dasac_conv_mpad(M), Kpad = dasac_conv_kpad(K);
  hipLaunchKernelGGL(pack_x3, dim3(stream_grid((int64_t)(Kpad / 8) * Mpad, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const f32x4*>(packed), reinterpret_cast<f32x4*>(packed_x3), Mpad, Kpad);
  DASAC_CHECK_LAUNCH("pack_x3");
  return DASAC_OK;
}

// Number of pixel splits: every block of a wgrad launch runs for the same time, so pick the split count
// whose block total fills whole rounds of resident blocks (2 per CU for the 128-row tile, 3 otherwise).
static int wgrad_bn(int Cx) { return (Cx % 128 != 0 && Cx % 64 == 0) ? 64 : 128; }   // k-tile rows: one tap per tile when possible

// 3 resident blocks per CU (<= 168 registers).  Measured in round 3: both 128x128 kernels also fit 128 registers (2 / 44 spill
// instructions outside the MFMA loop) and 4 x 36 KB of LDS, but at 4 blocks per CU the step's weight-gradient time goes from
// 109.2 to 114.0 ms (123.1 -> 117.9 TFLOP/s): occupancy is not what the pixel loop lacks.  Nor is it the dZ loads of the 3x3
// layers: dZ through the four-pixels-per-lane loader (dwordx4 + ds_write_b128, the gathered operand unchanged) gives 109.6 vs
// 109.1 ms.
static int wgrad_splits(int Mpad, int Kpad, int Npix, int BM, int BNk = 128) {
  const int tiles = (Mpad / BM) * (Kpad / BNk);
  const int slots = kNumCu * 3;
  // At least 256 pixels (8 K-steps) per split.  Rounds 1-3 asked for 1024: at batch 2 (cfg-2: 18 818 pixels) that capped the 16-tile
  // 1x1 layers at 19 splits = 304 blocks for 768 slots; measured in round 4 (profiles/r4_wgrad_split_granularity.txt, cfg-2):
  // 1024 -> 512 -> 256 pixels: weight gradients 18.7 -> 16.3 -> 15.4 ms/step (90 -> 103 -> 109 TFLOP/s).  Large batches are
  // unaffected (the cap of 64 splits binds first).
  constexpr int kMinPix = 256;
  int max_splits = (Npix + kMinPix - 1) / kMinPix;
  // few tiles x many pixels (layer1 / stem: 2 tiles, 298k..1.2M pixels): more splits, or 128 blocks would face 768 slots
  const int cap = (tiles < 12 && Npix >= 200000) ? (slots + tiles - 1) / tiles : 64;   // measured: no gain at 97x97 resolution
  if (max_splits > cap) max_splits = cap;
  if (max_splits < 1) max_splits = 1;
  int best = 1;
  double best_eff = 0.0;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const int blocks = tiles * sp;
    const int rounds = (blocks + slots - 1) / slots;
    const double eff = (double)blocks / ((double)rounds * slots);
    if (eff > best_eff + 0.02) {                             // prefer fewer splits unless clearly better
      best_eff = eff;
      best = sp;
    }
  }
  return best;
}

extern "C" size_t dasac_conv_wgrad_workspace(int Nb, int OH, int OW, int M, int K) {
  const int Mpad = dasac_conv_mpad(M), Kpad = dasac_conv_kpad(K);
  // upper bound over both k-tile widths (the width depends on Cx, unknown here)
  const int s128 = wgrad_splits(Mpad, Kpad, Nb * OH * OW, pick_bm(Mpad), 128), s64 = wgrad_splits(Mpad, Kpad, Nb * OH * OW, pick_bm(Mpad), 64);
  const int splits = s128 > s64 ? s128 : s64;
  return (size_t)splits * Mpad * (Kpad + 1) * sizeof(float);     // slabs + per-split channel sums
}

// dz [Nb,M,OH,OW], x [Nb,Cx,H,W] -> slabs in workspace -> dW (one or several weight tensors of a fused conv)
static int conv_wgrad_impl(bool x3, const float* dz, const float* x, const int32_t* table, int Nb, int Cx, int H, int W, int OH,
                           int OW, int stride, int M, int K, void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(dz && x && table && workspace, "conv_wgrad: null pointer");
  GemmGeom g;
  const int Mpad = dasac_conv_mpad(M), Kpad = dasac_conv_kpad(K);
  int rc = fill_geom(g, Nb, Cx, H, W, OH, OW, stride, M, Mpad, Kpad, OH, OW, 1);
  if (rc) return rc;
  const int bm = pick_bm(Mpad);
  const int bnk = (M % 8 == 0 && bm != 32) ? wgrad_bn(Cx) : 128;
  const int splits = wgrad_splits(Mpad, Kpad, g.Npix, bm, bnk);
  if (ws_bytes < (size_t)splits * Mpad * (Kpad + 1) * sizeof(float)) return fail(DASAC_EWORKSPACE, "conv_wgrad: workspace too small");
  int per = (g.Npix + splits - 1) / splits;
  per = (per + kWgPix - 1) / kWgPix * kWgPix;
  const int4* tab = reinterpret_cast<const int4*>(table);
  float* P = reinterpret_cast<float*>(workspace);
  float* Psum = P + (size_t)splits * Mpad * Kpad;
  hipStream_t s = as_stream(stream);
  const bool fast = (Cx % 128 == 0) && (M % 8 == 0);   // a 128-row k tile sits inside one tap
  if (bnk == 64) {                                     // Cx = 64, 192, ...: 64-row k tiles, still one tap each
    if (bm == 128) launch_wgrad<128, 64, 2, true>(x3, dz, x, tab, P, Psum, g, splits, per, s);
    else launch_wgrad<64, 64, 2, true>(x3, dz, x, tab, P, Psum, g, splits, per, s);
    DASAC_CHECK_LAUNCH("conv_wgrad");
    return DASAC_OK;
  }
  switch (bm) {
    case 128: {
      // 1x1 stride-1 layers: both operands are plain [row][pixel] matrices -> four pixels per lane (conv_wgrad<..., QUAD>)
      const bool same = fast && !x3 && stride == 1 && H == OH && W == OW && M % 128 == 0;
      const bool quad = same && K == Cx;                                                   // one tap, no padding
      // "same"-padded k x k layers (every 3x3 of the bottlenecks): the quad loader with per-pixel tap validity (QTAP)
      const bool qtap = same && K > Cx && K % Cx == 0 && W >= 4;
      if (quad || qtap) {
        const int m_tiles = g.Mpad / 128, k_tiles = g.Kpad / 128;
        const int grid = (m_tiles * k_tiles * splits + kNumXcd - 1) / kNumXcd * kNumXcd;
        if (quad)
          hipLaunchKernelGGL((conv_wgrad<128, 128, 2, true, false, true, false>), dim3(grid), dim3(kThreads), 0, s, dz, x, tab, P, Psum, g,
                             m_tiles, k_tiles, splits, per);
        else
          hipLaunchKernelGGL((conv_wgrad<128, 128, 2, true, false, true, true>), dim3(grid), dim3(kThreads), 0, s, dz, x, tab, P, Psum, g,
                             m_tiles, k_tiles, splits, per);
      } else if (fast) launch_wgrad<128, 128, 2, true>(x3, dz, x, tab, P, Psum, g, splits, per, s);
      else launch_wgrad<128, 128, 2, false>(x3, dz, x, tab, P, Psum, g, splits, per, s);
      break;
    }
    case 64:
      if (fast) launch_wgrad<64, 128, 2, true>(x3, dz, x, tab, P, Psum, g, splits, per, s);
      else launch_wgrad<64, 128, 2, false>(x3, dz, x, tab, P, Psum, g, splits, per, s);
      break;
    default: launch_wgrad<32, 128, 1, false>(x3, dz, x, tab, P, Psum, g, splits, per, s); break;
  }
  DASAC_CHECK_LAUNCH("conv_wgrad");
  return DASAC_OK;
}

extern "C" int dasac_conv_wgrad(const float* dz, const float* x, const int32_t* table, int Nb, int Cx, int H, int W, int OH,
                                int OW, int stride, int M, int K, void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  return conv_wgrad_impl(false, dz, x, table, Nb, Cx, H, W, OH, OW, stride, M, K, workspace, ws_bytes, stream);
}

extern "C" int dasac_conv_wgrad_x3(const float* dz, const float* x, const int32_t* table, int Nb, int Cx, int H, int W, int OH,
                                   int OW, int stride, int M, int K, void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  return conv_wgrad_impl(true, dz, x, table, Nb, Cx, H, W, OH, OW, stride, M, K, workspace, ws_bytes, stream);
}

extern "C" int dasac_conv_wgrad_dot_rows(int Cin, int taps) { return Cin < 32 ? (Cin * taps + 63) / 64 : (Cin + 63) / 64; }

extern "C" int dasac_conv_wgrad_finish(const void* workspace, int Nb, int OH, int OW, int M, int K, const float* w,
                                       const float* scale, float* dw, float* dot, float* sum_dz, int Cin, int taps,
                                       int tap0, dasac_stream_t stream) {
  DASAC_REQUIRE(workspace && w && dw, "conv_wgrad_finish: null pointer");
  DASAC_REQUIRE(!dot || (tap0 == 0 && Cin * taps == K), "conv_wgrad_finish: the dot term is defined for single-branch convolutions");
  const int Mpad = dasac_conv_mpad(M), Kpad = dasac_conv_kpad(K);
  const int bm = pick_bm(Mpad);
  const int splits = wgrad_splits(Mpad, Kpad, Nb * OH * OW, bm, (M % 8 == 0 && bm != 32) ? wgrad_bn(Cin) : 128);
  const float* P = reinterpret_cast<const float*>(workspace);
  const float* Psum = P + (size_t)splits * Mpad * Kpad;
  if (Cin < 32)      // lanes along the whole k axis (see the kernel); tap0 * Cin is this branch's first k
    hipLaunchKernelGGL(wgrad_reduce_scalar, dim3((Cin * taps + 63) / 64, M), dim3(256), 0, as_stream(stream), P + (size_t)tap0 * Cin, Psum,
                       splits, Mpad, Kpad, w, scale, dw, dot, sum_dz, Cin * taps, 1, 0, Cin);
  else if (Cin % 64 != 0)
    hipLaunchKernelGGL(wgrad_reduce_scalar, dim3((Cin + 63) / 64, M), dim3(256), 0, as_stream(stream), P, Psum, splits, Mpad, Kpad, w,
                       scale, dw, dot, sum_dz, Cin, taps, tap0, 0);
  else if (taps == 1 && tap0 == 0 && K == Cin)
    hipLaunchKernelGGL(wgrad_reduce_1x1, dim3((Cin + 255) / 256, M), dim3(256), 0, as_stream(stream), P, Psum, splits, Mpad, Kpad, w, scale,
                       dw, dot, sum_dz, Cin);
  else
    hipLaunchKernelGGL(wgrad_reduce, dim3(Cin / 64, M), dim3(256), 0, as_stream(stream), P, Psum, splits, Mpad, Kpad, w, scale, dw, dot,
                       sum_dz, Cin, taps, tap0);
  DASAC_CHECK_LAUNCH("wgrad_reduce");
  return DASAC_OK;
}

// ------------------------------------------------------------------------------------------------
// "Tap-expanded" evaluation of few-output-channel, many-tap convolutions (the ASPP classifiers:
// 4 x (3x3, dilation 6..24), 2048|1024 -> 19, deeplabv2.py:101-116).  A 19-row GEMM wastes 40 % of a
// 32-row MFMA tile and re-reads the activations once per tap; instead
//    forward :  Y[(tap,co)][p] = sum_ci W[co,ci,tap] x[ci][p]      -- ONE dense 1x1 GEMM, M = taps*Cp
//               out[co][p]     = bias[co] + sum_tap Y[(tap,co)][p + shift(tap)]            (tap_gather)
//    backward:  D[(tap,co)][p'] = dout[co][p' - shift(tap)]                                 (tap_scatter)
//               dW = 1x1 wgrad(D, x),  dx = 1x1 dgrad(D)  -- again dense GEMMs over taps*Cp channels.
// Cp >= Cout pads the channel count so that taps*Cp is a multiple of 16 (FAST path of conv_gemm).
// ------------------------------------------------------------------------------------------------
namespace dasac {

struct TapShifts {
  int n;
  short dh[64], dw[64];
};

// transposed = 0: k = ci, m = (tap0+tap)*Cp + co ;  transposed = 1: k = (tap0+tap)*Cp + co, m = ci
__global__ void pack_expanded(const float* __restrict__ Wt, float* __restrict__ Wp, int Cout, int Cin, int taps, int tap0,
                              int Cp, int Mpad, int transposed) {
  const int64_t total = (int64_t)Cout * Cin * taps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const int64_t r = i / taps;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    const int e = (tap0 + tap) * Cp + co;
    const int64_t k = transposed ? e : ci;
    const int m = transposed ? ci : e;
    Wp[((k >> 2) * Mpad + m) * 4 + (k & 3)] = Wt[i];
  }
}

__global__ __launch_bounds__(256) void tap_gather(const float* __restrict__ Y, TapShifts ts, int Cp, int Cout,
                                                  const float* __restrict__ bias, int H, int W, float* __restrict__ out,
                                                  int64_t total) {
  const int HW = H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const int64_t r = i / HW;
    const int co = (int)(r % Cout);
    const int64_t b = r / Cout;
    const int oh = p / W, ow = p - oh * W;
    const float* yb = Y + (size_t)b * ts.n * Cp * HW;
    float acc = bias ? bias[co] : 0.f;
    for (int t = 0; t < ts.n; ++t) {
      const int ih = oh + ts.dh[t], iw = ow + ts.dw[t];
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) acc += yb[(size_t)(t * Cp + co) * HW + ih * W + iw];
    }
    out[i] = acc;
  }
}

__global__ __launch_bounds__(256) void tap_scatter(const float* __restrict__ dout, TapShifts ts, int Cp, int Cout, int H,
                                                   int W, float* __restrict__ D, int64_t total) {
  const int HW = H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const int64_t r = i / HW;
    const int e = (int)(r % (ts.n * Cp));
    const int64_t b = r / (ts.n * Cp);
    const int t = e / Cp, co = e - t * Cp;
    const int ih = p / W - ts.dh[t], iw = p % W - ts.dw[t];
    float v = 0.f;
    if (co < Cout && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
      v = dout[((size_t)b * Cout + co) * HW + ih * W + iw];
    D[i] = v;
  }
}

// dW[co][ci][tap] = sum_s P[s][(tap0+tap)*Cp + co][ci]
__global__ __launch_bounds__(256) void wgrad_reduce_expanded(const float* __restrict__ P, int splits, int Mpad, int Kpad,
                                                             float* __restrict__ dW, int Cout, int Cin, int taps, int tap0,
                                                             int Cp) {
  const int64_t total = (int64_t)Cout * taps * Cin;
  const size_t slab = (size_t)Mpad * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % Cin);                       // slab reads coalesce along ci
    const int64_t r = i / Cin;
    const int tap = (int)(r % taps), co = (int)(r / taps);
    const size_t idx = (size_t)((tap0 + tap) * Cp + co) * Kpad + ci;
    dW[((size_t)co * Cin + ci) * taps + tap] = sum_splits(P + idx, slab, splits);
  }
}

static int fill_shifts(TapShifts& ts, const int32_t* kh, const int32_t* kw, const int32_t* dil, const int32_t* pad, int nb) {
  ts.n = 0;
  for (int b = 0; b < nb; ++b)
    for (int a = 0; a < kh[b]; ++a)
      for (int c = 0; c < kw[b]; ++c) {
        if (ts.n >= 64) return fail(DASAC_EINVAL, "tap-expanded conv: more than 64 taps");
        ts.dh[ts.n] = (short)(a * dil[b] - pad[b]);
        ts.dw[ts.n] = (short)(c * dil[b] - pad[b]);
        ++ts.n;
      }
  return DASAC_OK;
}

}  // namespace dasac

extern "C" int dasac_conv_pack_expanded(const float* w, int Cout, int Cin, int taps, int tap0, int total_taps, int Cp,
                                        int transposed, float* packed, dasac_stream_t stream) {
  DASAC_REQUIRE(w && packed && Cp >= Cout, "conv_pack_expanded: bad arguments");
  const int E = total_taps * Cp;
  const int M = transposed ? Cin : E, K = transposed ? E : Cin;
  const int Mpad = dasac_conv_mpad(M), Kpad = dasac_conv_kpad(K);
  hipStream_t s = as_stream(stream);
  if (tap0 == 0) DASAC_HIP(hipMemsetAsync(packed, 0, (size_t)Kpad * Mpad * sizeof(float), s));
  hipLaunchKernelGGL(pack_expanded, dim3(stream_grid((int64_t)Cout * Cin * taps, 256)), dim3(256), 0, s, w, packed, Cout, Cin, taps,
                     tap0, Cp, Mpad, transposed);
  DASAC_CHECK_LAUNCH("pack_expanded");
  return DASAC_OK;
}

extern "C" int dasac_tap_gather(const float* y, const int32_t* kh, const int32_t* kw, const int32_t* dil, const int32_t* pad,
                                int n_branches, int Cp, int Cout, const float* bias, int B, int H, int W, float* out,
                                dasac_stream_t stream) {
  DASAC_REQUIRE(y && out && kh && kw && dil && pad, "tap_gather: null pointer");
  TapShifts ts;
  int rc = fill_shifts(ts, kh, kw, dil, pad, n_branches);
  if (rc) return rc;
  const int64_t total = (int64_t)B * Cout * H * W;
  hipLaunchKernelGGL(tap_gather, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), y, ts, Cp, Cout, bias, H, W, out, total);
  DASAC_CHECK_LAUNCH("tap_gather");
  return DASAC_OK;
}

extern "C" int dasac_tap_scatter(const float* dout, const int32_t* kh, const int32_t* kw, const int32_t* dil,
                                 const int32_t* pad, int n_branches, int Cp, int Cout, int B, int H, int W, float* d,
                                 dasac_stream_t stream) {
  DASAC_REQUIRE(dout && d && kh && kw && dil && pad, "tap_scatter: null pointer");
  TapShifts ts;
  int rc = fill_shifts(ts, kh, kw, dil, pad, n_branches);
  if (rc) return rc;
  const int64_t total = (int64_t)B * ts.n * Cp * H * W;
  hipLaunchKernelGGL(tap_scatter, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), dout, ts, Cp, Cout, H, W, d, total);
  DASAC_CHECK_LAUNCH("tap_scatter");
  return DASAC_OK;
}

extern "C" int dasac_conv_wgrad_finish_expanded(const void* workspace, int Nb, int OH, int OW, int E, int Cin, float* dw,
                                                int Cout, int taps, int tap0, int Cp, dasac_stream_t stream) {
  DASAC_REQUIRE(workspace && dw, "conv_wgrad_finish_expanded: null pointer");
  const int Mpad = dasac_conv_mpad(E), Kpad = dasac_conv_kpad(Cin);
  const int bm = pick_bm(Mpad);
  const int splits = wgrad_splits(Mpad, Kpad, Nb * OH * OW, bm, (E % 8 == 0 && bm != 32) ? wgrad_bn(Cin) : 128);
  hipLaunchKernelGGL(wgrad_reduce_expanded, dim3(stream_grid((int64_t)Cout * taps * Cin, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float*>(workspace), splits, Mpad, Kpad, dw, Cout, Cin, taps, tap0, Cp);
  DASAC_CHECK_LAUNCH("wgrad_reduce_expanded");
  return DASAC_OK;
}

