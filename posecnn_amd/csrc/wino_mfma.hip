// wino_mfma.hip — the Winograd-domain contractions of the VGG16 trunk on the gfx950 matrix cores,
// with the output transform fused in (`Network.conv` for every 3x3 / stride 1 / SAME layer with
// Cin, Cout multiples of 64: lib/networks/network.py:159-187, lib/networks/vgg16_convs.py:37-52 and
// the depth tower :54-67). Replaces the library batched GEMM + wino43_output_kernel pair of round 1
// and its Cin = 64-only fused kernel.
//
//   y = [ReLU](A^T (sum_ci V[k][t][ci] U[k][ci][co]) A + bias[co])   [2x2 max-pooled]
//
// F(4x4,3x3): 36 planes k = 6 xi + nu, each a GEMM [T x Cin] . [Cin x Cout] in exact f32 on
// v_mfma_f32_16x16x4_f32 (no xf32/TF32 on gfx950; 157 TFLOP/s peak). Design for the chip, not for a
// generic GEMM:
//
//  * Workgroup = 32 tiles x 64 output channels, 4 waves, TWO workgroups per CU (one wave of each per
//    SIMD, so that one workgroup's LDS / global / VALU work, barrier waits and epilogue hide under the
//    other's MFMAs; the 64-tile / 8-wave / one-per-CU shape, WR = 2, measures 2-7 % slower and is kept
//    for the ablation harness). Wave wn owns the block's 32 tiles x channels [16 wn, 16 wn + 16): two
//    16x16 accumulator blocks.
//  * The transform-domain product M NEVER exists, not even in registers as a whole: planes are
//    processed column by column (nu outer, xi inner). Once the 6 planes of column nu are summed
//    over all of Cin, t = A^T M[:, nu] (4 values) is folded into the 16 outputs Y += t (x) A[nu, :]
//    and the 6 accumulators are recycled. 22 live values per (tile, channel) instead of 36 is what
//    lets 64 x 64 elements per CU (two 32 x 64 blocks) fit the register file (360 KB of the CU's
//    512 KB) — 4x the tile of the round-1 kernel.
//  * Operands go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no
//    ds_write pass) through a ring of 3 stage buffers, ONE barrier per 64-deep K stage: the 4 DMA
//    instructions a wave issues for stage s+2 fly under the MFMAs of stages s and s+1 (a stage of a
//    Cin = 64 layer is a fresh HBM miss per plane: one stage of cover is not enough). The barrier
//    is a raw s_barrier behind a COUNTED s_waitcnt vmcnt(4) — __syncthreads() would drain the DMA
//    queue (vmcnt(0)) every stage. The LDS image is the DMA's
//    lane-linear one (rows of 64 floats, unpadded); bank conflicts are avoided by an XOR swizzle of
//    the 16-byte chunk index with the row (applied to the SOURCE address of the DMA and to the
//    ds_read_b128 address — the same involution on both sides). One b128 read feeds 4 MFMAs (K is
//    consumed in the order 16 j + 4 (lane >> 4) + i for both operands).
//  * Epilogue: bias + ReLU (+ 2x2 max-pool of the 4x4 tile) in registers, then through LDS (two staging
//    buffers, one barrier per pass) so that every store instruction writes whole 256-byte channel rows
//    (the C/D layout alone would give 64-byte segments).
//  * Launches too small to fill the chip (batch-1 conv4_x / conv5_x) split Cin over grid.y; a reduction
//    kernel sums the raw partial outputs in a fixed order and applies bias / ReLU / pooling.
//  * blockIdx -> (tile block, channel block) is XCD-aware: a tile block's V rows are shared by the
//    Cout / 64 workgroups that run back to back on ONE XCD (same L2), tile blocks are dealt round
//    robin to the 8 XCDs.
//  * `groups`: G independent filter sets over G equal slices of the tile range — the colour and the
//    depth tower of an RGB-D network (identical shapes, different weights) run as ONE launch with
//    twice the workgroups.
#include <cstdlib>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WM_BC = 64;     // output channels per workgroup
constexpr int WM_KC = 64;     // K (input channels) per pipeline stage
constexpr int WM_LD = 64;     // LDS row (floats): unpadded, XOR-swizzled 16-byte chunks
constexpr int WM_NBUF = 3;    // stage buffers in the LDS ring (prefetch distance 2)

// 16 bytes per lane, global -> LDS, no register round trip. LDS destination = wave-uniform base +
// 16 * lane (so the image is lane-linear); the per-lane SOURCE address carries the swizzle.
__device__ __forceinline__ void glds16(const char* g, float* lds_wave_base)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same DMA with the addressing spelled out: SGPR base + 32-bit VGPR byte offset, LDS base in M0. Through
// the builtin the compiler turned most of these into 64-bit VGPR addresses (a v_mov_b64 + v_lshl_add_u64 pair
// per load, ~20 VALU instructions per stage next to 32 MFMAs); here the per-stage pointer arithmetic stays on
// the scalar unit. Completion is tracked by the hand-placed s_waitcnt vmcnt(N) of the stage barrier.
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned lds_addr)
{
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// Output transform, one column of the 6x6 transform domain at a time:
//   t = A^T m   with A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6_col(const float* m, float* t)
{
  const float s = m[1] + m[2], d = m[1] - m[2], S = m[3] + m[4], D = m[3] - m[4];
  t[0] = (m[0] + s) + S;
  t[1] = __builtin_fmaf(2.f, D, d);     // (explicit FMAs: this file is built -ffp-contract=off; the fold is 12 %
  t[2] = __builtin_fmaf(4.f, S, s);     //  of a Cin = 64 layer, and one instruction per term instead of two
  t[3] = __builtin_fmaf(8.f, D, d) + m[5];   // buys 1.7 % of the 12-layer trunk)
}

// POOL: 0 = y [.,H,W,Cout]; 1 = only max_pool_2x2(y) (written to y); 2 = both (y and ypool)
// WR: wave rows — 1 (what the library launches): 32 tiles x 64 channels, 4 waves, two workgroups per CU;
// 2: 64 x 64, 8 waves, one workgroup per CU (kept for the ablation harness; see the launcher for the numbers).
// ABL (tools/wino_ablate.hip only; the library instantiates 0): leave one ingredient of the K loop out to
// see what it costs — 1 no barrier, 2 no DMA, 4 no LDS reads, 8 no MFMAs, 16 no column fold, 32 no epilogue, 64 every
// workgroup on the operands of block (0, 0) (L2-resident: what the fabric costs).
// ZC (round 4): the first MFMA pair into a plane's accumulators takes C = 0 as an inline constant instead of reading
// registers that a v_mov zeroed after every column fold (48 VALU per fold; on the Cin = 64 layers a fold comes every 6
// stages and VALU time is paid in full next to the MFMAs). Same bits: 0 + a b either way.
// `mode` bit 0: XCD x owns channel block x (launches with ncb == 8 whose filter bank outweighs their transformed input,
// i.e. the deep layers of a single frame: every XCD then streams ITS eighth of U once instead of all of U).
// Measured and dropped in round 4 (tools/archive/r4_wino_modes.sh, 12-layer totals at 2 x 16 frames):
//   * s_setprio 3 / 0 by the workgroup's slot on the CU (HW_ID.TG_ID), so that one of the two co-resident workgroups runs
//     as if alone and the other fills the matrix pipe's gaps: 14.62 -> 14.71 / 14.90 ms for the two polarities — the
//     symmetric pair is the better schedule;
//   * persistent workgroups (a grid of the 512 resident slots, each walking its (tile block, channel block) pairs, so that
//     the early layers' 36-72-stage pairs stop paying a dispatch, a cold prologue and a store drain each): 14.62 -> 15.07 ms
//     (conv4_2 1.61 -> 1.73, conv2_2 1.74 -> 1.83) — the hardware's dynamic hand-out of pairs to whichever slot frees first
//     balances better than a static walk, and the 8 extra live registers of the loop cost the kernel its last slack.
// EPI (round 5; not instantiated by the library — tools/wino_ablate.hip can): 1 = the MFMA operand roles swapped (A = U^T rows,
// B = V rows: the same products in the same K order, so the same bits — verified against EPI = 0 on seven layer shapes, all pool
// modes, both block maps) — a lane then holds FOUR CONSECUTIVE CHANNELS of one tile instead of one channel of four tiles, and the
// epilogue stores its outputs straight from registers as 16-byte pieces (16 lanes x 4 quads = a tile's 64 channels per 256
// bytes): no LDS staging, no barriers. The warm ablation had priced the staging at 7 / 5 / 4 % of conv2_1 / conv2_2 / conv3_2;
// measured, the direct form gives it back in 64-byte partial-line stores: 12 layers 14.136 vs 14.169 ms alone (conv2_1 1.052
// vs 1.022, conv2_2 1.656 vs 1.676, conv3_1 0.868 vs 0.885, conv4_2 1.569 vs 1.589), the kernel inside the step 1107 / 1118 vs
// 1098 / 1098 us per launch, the step 796 / 794 vs 764 / 808 frames/s: no gain — the staged epilogue (0) stays.
template <int POOL, int WR, int ABL = 0, int ZC = 1, int EPI = 0>
__global__ __launch_bounds__(256 * WR, 2) void wino43_mfma_kernel(
    const float* __restrict__ v, const float* __restrict__ ut, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ ypool, int H, int W, int Cin, int Cout, int Ht, int Wt,
    long long T, long long tiles_per_group, int relu, int nbt, int ncb, int ksplit, long long ysplit_stride, int mode)
{
  constexpr int BT = 32 * WR, NW = 4 * WR, NT = 256 * WR;
  constexpr int RING = WM_NBUF * (BT + 64) * WM_LD, STAGE2 = 2 * BT * 4 * 64;
  __shared__ __attribute__((aligned(16))) float smem[RING > STAGE2 ? RING : STAGE2];   // sA[3][BT][64] | sB[3][64][64]; epilogue: 2 x [BT][4][64]
  float(*sA)[BT][WM_LD] = reinterpret_cast<float(*)[BT][WM_LD]>(smem);
  float(*sB)[64][WM_LD] = reinterpret_cast<float(*)[64][WM_LD]>(smem + WM_NBUF * BT * WM_LD);

  // XCD-aware block map: XCD x = blockIdx % 8 takes tile blocks tb == x (mod 8); on an XCD the
  // channel blocks of a tile block are consecutive
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const bool cbmajor = (mode & 1) != 0;          // (ncb == 8, checked by the launcher)
  const int cb = cbmajor ? x : q % ncb;
  const int tb = cbmajor ? q : (q / ncb) * 8 + x;
  if (tb >= nbt) return;
  const int tb_data = (ABL & 64) ? 0 : tb, cb_data = (ABL & 64) ? 0 : cb;   // ABL 64 (timing only): every workgroup reads block (0, 0): all operands L2-resident
  y += (long long)blockIdx.y * ysplit_stride;    // `ksplit` > 1: this Cin slice's partial output (see below)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything derived from it (LDS bases, M0) stays on the SALU
  const int wm = wave & (WR - 1), wn = wave / WR;
  const int lr = lane & 15, lk = lane >> 4;
  // tile blocks never straddle a group: group g owns tiles [g tpg, (g+1) tpg) in nbg blocks of 64
  const int nbg = (int)((tiles_per_group + BT - 1) / BT);
  const int grp = tb / nbg;
  const long long t0 = (long long)grp * tiles_per_group + (long long)(tb_data - (ABL & 64 ? 0 : grp * nbg)) * BT;
  const long long tend = (long long)(grp + 1) * tiles_per_group;   // first tile that is not this block's business
  const float* utg = ut + (size_t)grp * 36 * Cout * Cin;
  // `ksplit` > 1 (small launches, see the launcher): workgroup blockIdx.y contracts only its slice of Cin and
  // writes a raw partial output (the host passes no bias, no ReLU, POOL = 0); a reduction kernel finishes
  const int ks = blockIdx.y;
  const int NK = Cin / WM_KC / ksplit;

  // staging: a stage = BT rows x 256 B of V and 64 rows of U^T = BT/4 + 16 DMA instructions of 4 rows
  // each; wave w issues V instructions 2w, 2w + 1 and U^T instructions (16/NW) w ... Lane l of
  // instruction ii lands at row 4 ii + (l >> 4), physical chunk l & 15, and therefore fetches logical
  // chunk (l & 15) ^ (row & 15).
  constexpr int UB = 16 / NW;                                   // U^T instructions per wave (2 or 4)
  const int dr0 = 8 * wave + lk, dr1 = dr0 + 4;                 // the lane's V rows in its two instructions
  const int dc0 = (lr ^ (dr0 & 15)) * 4, dc1 = (lr ^ (dr1 & 15)) * 4;
  const int ur0 = 4 * UB * wave + lk;                           // first U^T row; instruction i adds 4 i
  const long long tlast = tend - 1;
  const long long ta0 = t0 + dr0 < tend ? t0 + dr0 : tlast, ta1 = t0 + dr1 < tend ? t0 + dr1 : tlast;   // rows past the end: any finite data, never stored
  // per-lane BYTE offsets (32 bit) from wave-uniform bases: the DMA then addresses as SGPR base + VGPR
  // offset and the per-stage pointer arithmetic stays on the scalar unit
  const unsigned va0 = (unsigned)((ta0 * Cin + dc0) * 4), va1 = (unsigned)((ta1 * Cin + dc1) * 4);
  unsigned ub[UB];
#pragma unroll
  for (int i = 0; i < UB; i++) ub[i] = (unsigned)((((size_t)(cb_data * WM_BC + ur0 + 4 * i)) * Cin + (lr ^ ((ur0 + 4 * i) & 15)) * 4) * 4);
  const char* vbase = reinterpret_cast<const char*>(v);
  const char* ubase = reinterpret_cast<const char*>(utg);
  const long long vplane = T * Cin;
  const long long uplane = (long long)Cout * Cin;
  const int ldsw = 8 * wave * WM_LD;                            // this wave's first V row (floats) inside a stage buffer
  const int ldsu = 4 * UB * wave * WM_LD;                       // ... and its first U^T row
  // LDS byte addresses of those rows in stage buffer 0 (wave-uniform: M0 of the DMA)
  const unsigned lds_base_ = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  const unsigned lds_a0 = lds_base_ + (unsigned)ldsw * 4u;
  const unsigned lds_b0 = lds_base_ + (unsigned)(WM_NBUF * BT * WM_LD + ldsu) * 4u;

  // the lane's bias value, requested HERE: issued where it is first used (the epilogue, round 4) it was a bare global
  // round trip between the last MFMA and the first staged row of every workgroup (~1 us of a 15-us conv2_1 workgroup)
  const float bv = (bias && !EPI) ? bias[(size_t)grp * Cout + cb * WM_BC + 16 * wn + lr] : 0.f;   // (no bias: the raw partial output of a Cin split)
  const v4f bv4 = (bias && EPI) ? *reinterpret_cast<const v4f*>(bias + (size_t)grp * Cout + cb * WM_BC + 16 * wn + 4 * lk) : (v4f){0.f, 0.f, 0.f, 0.f};
  v4f acc[6][2];
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i][0] = acc[i][1] = (v4f){0.f, 0.f, 0.f, 0.f};
  // outputs of the lane's elements: [block][pair of C/D rows i][4 a + e][i & 1] — rows i, i + 1 side by side, so that the fold
  // (PKF) can work on register PAIRS as the MFMA delivers them (acc[x][b] = rows 0..3 in four consecutive registers)
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f yo2[2][2][16];
#define YO(B_, I_, O_) yo2[B_][(I_) >> 1][O_][(I_) & 1]
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int o = 0; o < 16; o++) yo2[b][i][o] = (v2f){0.f, 0.f};

#define WM_DMA(BUF, VO, UO)                                                  \
  if constexpr (!(ABL & 2)) {                                                \
    const char* vs_ = vbase + (VO) * 4;                                      \
    const char* us_ = ubase + (UO) * 4;                                      \
    const unsigned la_ = lds_a0 + (unsigned)(BUF) * (BT * WM_LD * 4);          \
    const unsigned lb_ = lds_b0 + (unsigned)(BUF) * (64 * WM_LD * 4);          \
    glds16_s(vs_, va0, la_);                                                 \
    glds16_s(vs_, va1, la_ + 4 * WM_LD * 4);                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < UB; i_++)                         \
      glds16_s(us_, ub[i_], lb_ + 4 * i_ * WM_LD * 4);                        \
  }

  // Stage order: nu outer, xi, then kc fastest; (pnu, pxi, pkc) is the prefetch pointer, two stages
  // ahead of the compute pointer. Prologue: stages 0 and 1 in flight.
  // The prefetch offsets (floats) advance incrementally on the scalar unit: next K slice +64; next
  // plane of the column k += 6; next column k: 30 + nu -> nu + 1.
  int pnu = 0, pxi = 0, pkc = 0;
  long long pvo = (long long)ks * NK * WM_KC, puo = pvo;   // this split's first input channel
  const long long kback = (long long)(NK - 1) * WM_KC;
  // uniform branches on the scalar unit (3-4 SALU instructions on the common path; a select-only
  // version cost ~40 per stage and measured 1.2 % slower over the 12 layers); past the last stage the
  // pointer parks on it (the two extra prefetches re-load the last stage into a free buffer and are
  // drained before the epilogue)
  const long long dvx = 6 * vplane - kback, dux = 6 * uplane - kback;       // next plane of the column
  const long long dvn = -29 * vplane - kback, dun = -29 * uplane - kback;   // next column
#define WM_PF_ADVANCE()                                                                          \
  do {                                                                                           \
    if (pkc + 1 < NK) { pkc++; pvo += WM_KC; puo += WM_KC; }            /* next K slice */        \
    else if (pxi < 5) { pkc = 0; pxi++; pvo += dvx; puo += dux; }       /* next plane of the column */ \
    else if (pnu < 5) { pkc = 0; pxi = 0; pnu++; pvo += dvn; puo += dun; }   /* next column */     \
  } while (0)   /* past the last stage the pointer parks on it */
  { WM_DMA(0, pvo, puo); WM_PF_ADVANCE(); }
  { WM_DMA(1, pvo, puo); WM_PF_ADVANCE(); }
  int cur = 0;

  // ds_read side of the swizzle: row R = (..) + lr, logical chunk 4 j + lk -> physical (4 j + lk) ^ lr
  const int ra0 = (32 * wm + lr) * WM_LD, rb = (16 * wn + lr) * WM_LD;   // (+ 16 rows for the second block)

  // The K loop is software pipelined by half a stage: the 16 MFMAs of K groups 2, 3 of stage s-1 are
  // issued AFTER the barrier of stage s, behind the LDS reads of groups 0, 1 of stage s — so the LDS
  // latency that follows every barrier (all 8 waves would otherwise wait for their first operands at
  // the same moment, with the matrix pipe idle) and the latency of the mid-stage reads are both
  // covered by MFMAs that are already queued. Two operand register sets: X (groups 0, 1), Y (2, 3).
  v4f xa0[2], xa1[2], xb[2], ya0[2], ya1[2], yb[2];
  // The LDS reads are inline asm: hipcc's own s_waitcnt insertion collapses to lgkmcnt(0) across the
  // loop back edge (it would wait for reads issued a moment ago), so the reads are invisible to it
  // and the waits are placed by hand: lgkmcnt(0) inside the stage barrier retires the Y reads, one
  // lgkmcnt(0) after Y's MFMAs retires the X reads. sched_barrier(0) pins the order around them.
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  unsigned adA[4], adB[4];   // byte addresses of this lane's operand chunks in stage buffer 0, per K group
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned chb = (unsigned)(((4 * j + lk) ^ lr) * 16);
    adA[j] = lds0 + (unsigned)ra0 * 4u + chb;
    adB[j] = lds0 + (unsigned)(WM_NBUF * BT * WM_LD + rb) * 4u + chb;
  }
#define WM_DSREAD(DST, ADDR, OFF)                                                                     \
  if constexpr (!(ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR)); \
  else asm volatile("" : "+v"(DST) : "v"(ADDR))
#define WM_READ(SA0, SA1, SB, G0)                                                                     \
  _Pragma("unroll") for (int g_ = 0; g_ < 2; g_++) {                                                  \
    const unsigned aa_ = adA[(G0) + g_] + curA, ab_ = adB[(G0) + g_] + curB;                           \
    WM_DSREAD(SB[g_], ab_, 0);                                                                        \
    WM_DSREAD(SA0[g_], aa_, 0);                                                                       \
    WM_DSREAD(SA1[g_], aa_, 4096);   /* 16 rows further */                                             \
  }
#define WM_MFMA1(SA0, SA1, SB, G, ACC)                                                                \
  if constexpr (ABL & 8) { asm volatile("" :: "v"(SA0[G]), "v"(SA1[G]), "v"(SB[G])); } else           \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {                                                  \
    ACC[0] = EPI ? __builtin_amdgcn_mfma_f32_16x16x4f32(SB[G][i_], SA0[G][i_], ACC[0], 0, 0, 0)       \
                 : __builtin_amdgcn_mfma_f32_16x16x4f32(SA0[G][i_], SB[G][i_], ACC[0], 0, 0, 0);      \
    ACC[1] = EPI ? __builtin_amdgcn_mfma_f32_16x16x4f32(SB[G][i_], SA1[G][i_], ACC[1], 0, 0, 0)       \
                 : __builtin_amdgcn_mfma_f32_16x16x4f32(SA1[G][i_], SB[G][i_], ACC[1], 0, 0, 0);      \
  }
#define WM_MFMA(SA0, SA1, SB, ACC)                                                                    \
  WM_MFMA1(SA0, SA1, SB, 0, ACC) WM_MFMA1(SA0, SA1, SB, 1, ACC)
  /* the same K group, but the first pair starts the accumulators from the constant 0 (first MFMAs of a plane) */
#define WM_MFMA1_Z(SA0, SA1, SB, G, ACC)                                                              \
  if constexpr (ABL & 8) { asm volatile("" :: "v"(SA0[G]), "v"(SA1[G]), "v"(SB[G])); } else {         \
    const v4f z_ = (v4f){0.f, 0.f, 0.f, 0.f};                                                         \
    ACC[0] = EPI ? __builtin_amdgcn_mfma_f32_16x16x4f32(SB[G][0], SA0[G][0], z_, 0, 0, 0)             \
                 : __builtin_amdgcn_mfma_f32_16x16x4f32(SA0[G][0], SB[G][0], z_, 0, 0, 0);            \
    ACC[1] = EPI ? __builtin_amdgcn_mfma_f32_16x16x4f32(SB[G][0], SA1[G][0], z_, 0, 0, 0)             \
                 : __builtin_amdgcn_mfma_f32_16x16x4f32(SA1[G][0], SB[G][0], z_, 0, 0, 0);            \
    _Pragma("unroll") for (int i_ = 1; i_ < 4; i_++) {                                                \
      ACC[0] = EPI ? __builtin_amdgcn_mfma_f32_16x16x4f32(SB[G][i_], SA0[G][i_], ACC[0], 0, 0, 0)     \
                   : __builtin_amdgcn_mfma_f32_16x16x4f32(SA0[G][i_], SB[G][i_], ACC[0], 0, 0, 0);    \
      ACC[1] = EPI ? __builtin_amdgcn_mfma_f32_16x16x4f32(SB[G][i_], SA1[G][i_], ACC[1], 0, 0, 0)     \
                   : __builtin_amdgcn_mfma_f32_16x16x4f32(SA1[G][i_], SB[G][i_], ACC[1], 0, 0, 0);    \
    }                                                                                                 \
  }
  // column NU of the transform domain is complete: t = A^T M[:, NU]; Y[a][e] += t[a] * A[NU][e]
  // with A[NU][:] = (1,0,0,0) (1,1,1,1) (1,-1,1,-1) (1,2,4,8) (1,-2,4,-8) (0,0,0,1); accumulators recycled
#define WM_FOLD(NU)                                                                                   \
  if constexpr (!(ABL & 16)) {                                                                        \
    const int n_ = (NU);                                                                              \
    const float c0 = n_ == 5 ? 0.f : 1.f;                                                             \
    const float c1 = n_ == 1 ? 1.f : n_ == 2 ? -1.f : n_ == 3 ? 2.f : n_ == 4 ? -2.f : 0.f;           \
    const float c2 = (n_ == 1 || n_ == 2) ? 1.f : (n_ == 3 || n_ == 4) ? 4.f : 0.f;                   \
    const float c3 = n_ == 1 ? 1.f : n_ == 2 ? -1.f : n_ == 3 ? 8.f : n_ == 4 ? -8.f : n_ == 5 ? 1.f : 0.f; \
    /* two C/D rows at a time on packed f32 (round 5): v_pk_add / v_pk_fma on the register pairs as the MFMA delivers them,  */ \
    /* element for element at6_col + the 16 rank-1 FMAs of rounds 2-4 (same bits; 12 layers 14.5 vs 14.7 ms alone, the    */ \
    /* step unchanged: 793 / 796 vs 792 / 794 frames/s)                                                                    */ \
    _Pragma("unroll") for (int b_ = 0; b_ < 2; b_++)                                                \
      _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) {                                            \
        v2f m_[6], t_[4];                                                                           \
        _Pragma("unroll") for (int x_ = 0; x_ < 6; x_++) m_[x_] = (v2f){acc[x_][b_][2 * h_], acc[x_][b_][2 * h_ + 1]}; \
        const v2f s_ = m_[1] + m_[2], d_ = m_[1] - m_[2], S_ = m_[3] + m_[4], D_ = m_[3] - m_[4];   \
        t_[0] = (m_[0] + s_) + S_;                                                                  \
        t_[1] = __builtin_elementwise_fma((v2f){2.f, 2.f}, D_, d_);                                 \
        t_[2] = __builtin_elementwise_fma((v2f){4.f, 4.f}, S_, s_);                                 \
        t_[3] = __builtin_elementwise_fma((v2f){8.f, 8.f}, D_, d_) + m_[5];                         \
        _Pragma("unroll") for (int a_ = 0; a_ < 4; a_++) {                                          \
          yo2[b_][h_][4 * a_ + 0] = __builtin_elementwise_fma(t_[a_], (v2f){c0, c0}, yo2[b_][h_][4 * a_ + 0]); \
          yo2[b_][h_][4 * a_ + 1] = __builtin_elementwise_fma(t_[a_], (v2f){c1, c1}, yo2[b_][h_][4 * a_ + 1]); \
          yo2[b_][h_][4 * a_ + 2] = __builtin_elementwise_fma(t_[a_], (v2f){c2, c2}, yo2[b_][h_][4 * a_ + 2]); \
          yo2[b_][h_][4 * a_ + 3] = __builtin_elementwise_fma(t_[a_], (v2f){c3, c3}, yo2[b_][h_][4 * a_ + 3]); \
        }                                                                                           \
      }                                                                                             \
    if constexpr (!ZC) {                                                                              \
      _Pragma("unroll") for (int x_ = 0; x_ < 6; x_++) acc[x_][0] = acc[x_][1] = (v4f){0.f, 0.f, 0.f, 0.f}; \
    }                                                                                                 \
  }

  // One stage on accumulator set XI (XP = the set of the stage before when that was another plane):
  //   own DMAs of this stage landed (counted wait: the next stage's stay in flight) -> barrier
  //   (everybody's landed, everybody done reading the previous stage) -> DMA of stage s+2 into the
  //   buffer the previous stage just freed -> reads X(s) -> MFMAs Y(s-1) -> reads Y(s) -> MFMAs X(s)
#define WM_STAGE(XI, XP)                                                                              \
  do {                                                                                                \
    if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                 \
    else if constexpr (UB == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");                     \
    const unsigned curA = (unsigned)cur * (BT * WM_LD * 4), curB = (unsigned)cur * (64 * WM_LD * 4);      \
    WM_READ(xa0, xa1, xb, 0);                                                                         \
    __builtin_amdgcn_sched_barrier(0);   /* the X reads go out first: their latency hides under Y's MFMAs */ \
    if (kc > 0) { WM_MFMA(ya0, ya1, yb, acc[XI]); }                                                   \
    else if ((XI) > 0) { WM_MFMA(ya0, ya1, yb, acc[XP]); }                                            \
    else if (nu > 0) { WM_MFMA(ya0, ya1, yb, acc[5]); WM_FOLD(nu - 1); }                              \
    {                                                                                                 \
      const int nb = cur >= 1 ? cur - 1 : 2;   /* (cur + 2) % 3: the previous stage's buffer */         \
      WM_DMA(nb, pvo, puo);                                                                           \
      WM_PF_ADVANCE();                                                                                \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the X reads (issued 16 MFMAs ago) */        \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (ZC && kc == 0) { WM_MFMA1_Z(xa0, xa1, xb, 0, acc[XI]); }   /* a plane's first MFMAs: C = 0 */   \
    else { WM_MFMA1(xa0, xa1, xb, 0, acc[XI]); }                                                      \
    __builtin_amdgcn_sched_barrier(0);   /* the Y reads go out between X's MFMA groups: nothing waits for them before the next barrier */ \
    WM_READ(ya0, ya1, yb, 2);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    WM_MFMA1(xa0, xa1, xb, 1, acc[XI]);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* Y landed before the compiler may touch its registers (loop-carried) */ \
    cur = cur == 2 ? 0 : cur + 1;                                                                     \
  } while (0)

  for (int nu = 0; nu < 6; nu++) {
    for (int kc = 0; kc < NK; kc++) WM_STAGE(0, 5);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(1, 0);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(2, 1);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(3, 2);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(4, 3);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(5, 4);
  }
  // drain: K groups 2, 3 of the last stage, the last column, and every LDS read before the buffers
  // are recycled for the epilogue
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  WM_MFMA(ya0, ya1, yb, acc[5]);
  WM_FOLD(5);
  // all waves' parked prefetch DMAs have landed before the ring becomes the epilogue's staging buffer: the DMAs are
  // inline asm, so the compiler's __syncthreads() is a bare s_barrier without the vmcnt(0) it would need (see
  // csrc/fc_mfma.hip: the same omission corrupted output rows there)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#undef WM_STAGE
#undef WM_READ
#undef WM_DSREAD
#undef WM_MFMA
#undef WM_MFMA1
#undef WM_MFMA1_Z
#undef WM_FOLD
#undef WM_DMA
#undef WM_PF_ADVANCE

  if constexpr (ABL & 256) {   // ablation: bias + ReLU on the registers, nothing staged or stored
    const float bv_ = bv;
    float k_ = 0.f;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int o = 0; o < 16; o++) { float val = YO(b, i, o) + bv_; if (relu) val = val > 0.f ? val : 0.f; k_ += val; }
    if (k_ == 12345.678f) y[tid] = k_;
    return;
  }
  if constexpr (ABL & 32) {   // keep the accumulators alive, store one value per lane
    float k_ = 0.f;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int o = 0; o < 16; o++) k_ += YO(b, i, o);
#pragma unroll
    for (int x_ = 0; x_ < 6; x_++) k_ += acc[x_][0][0] + acc[x_][1][1];
    if (k_ == 12345.678f) y[tid] = k_;
    return;
  }
  if constexpr (EPI == 1) {
    // ---- epilogue, straight from registers (operand roles swapped: see EPI) ------------------------------------------------
    // lane holds, per block b, tile 32 wm + 16 b + lr x channels 16 wn + 4 lk + i (i = 0..3): YO(b, i, 4 a + e)
    static_assert(WR == 1 || !EPI, "the direct epilogue is written for 32-tile blocks");
    const int HtWt = Ht * Wt;
    const int bimg0 = (int)(t0 / HtWt);
    const int rem0 = (int)(t0 - (long long)bimg0 * HtWt);
    const int ty0 = rem0 / Wt, tx0 = rem0 - ty0 * Wt;
    const int co4 = cb * WM_BC + 16 * wn + 4 * lk;
    const int Hp = H / 2, Wp = W / 2;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int tl = 16 * b + lr;
      const unsigned n = (unsigned)(tx0 + tl), q = n / (unsigned)Wt, tx = n - q * (unsigned)Wt;
      const unsigned m = (unsigned)ty0 + q, q2 = m / (unsigned)Ht, ty = m - q2 * (unsigned)Ht;
      const int img = bimg0 + (int)q2;
      const bool mine = t0 + tl < tend;                      // (rows past the end of the group: computed on clamped operands, never stored)
      v4f o[16];
#pragma unroll
      for (int px = 0; px < 16; px++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float val = YO(b, i, px) + bv4[i];
          if (relu) val = val > 0.f ? val : 0.f;
          o[px][i] = val;
        }
      }
      if (POOL != 1) {
        float* yb = y + (((long long)img * H + 4 * (int)ty) * W + 4 * (int)tx) * Cout + co4;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (mine && 4 * (int)ty + a < H && 4 * (int)tx + e < W)
              *reinterpret_cast<v4f*>(yb + ((long long)a * W + e) * Cout) = o[4 * a + e];
      }
      if (POOL != 0) {
        float* yp = (POOL == 1 ? y : ypool) + (((long long)img * Hp + 2 * (int)ty) * Wp + 2 * (int)tx) * Cout + co4;
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
          for (int e2 = 0; e2 < 2; e2++) {
            v4f p = o[4 * (2 * a2) + 2 * e2];
            const v4f p1 = o[4 * (2 * a2) + 2 * e2 + 1], p2 = o[4 * (2 * a2 + 1) + 2 * e2], p3 = o[4 * (2 * a2 + 1) + 2 * e2 + 1];
#pragma unroll
            for (int i = 0; i < 4; i++) {
              p[i] = p1[i] > p[i] ? p1[i] : p[i];
              p[i] = p2[i] > p[i] ? p2[i] : p[i];
              p[i] = p3[i] > p[i] ? p3[i] : p[i];
            }
            if (mine && 2 * (int)ty + a2 < Hp && 2 * (int)tx + e2 < Wp)
              *reinterpret_cast<v4f*>(yp + ((long long)a2 * Wp + e2) * Cout) = p;
          }
      }
    }
    return;
  }
  // ---- epilogue -------------------------------------------------------------------------------
  // lane holds, per block b, tiles 32 wm + 16 b + 4 lk + i (i = 0..3) x channel 16 wn + lr
  const int col = 16 * wn + lr;
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int o = 0; o < 16; o++) {
        float val = YO(b, i, o) + bv;
        if (relu) val = val > 0.f ? val : 0.f;
        YO(b, i, o) = val;
      }

  // Staging through LDS: [BT tiles][4][64 channels] floats per pass, two buffers (the ring is free
  // now) so that a pass needs ONE barrier: pass a+1 writes the buffer whose readers all arrived at
  // barrier a. Writer lanes (lr, lk) hit rows 4 lk tiles apart = the same banks, so the 16-channel
  // group is XORed with lk (reader: same XOR, wave-uniform, float4 alignment kept).
  constexpr int SYF = BT * 4 * 64;
  const int wcol = col ^ (16 * lk);
  // The thread's 8 store slots: tile wave + NW r, pixel column re, channels rc4..rc4+3. Tile
  // coordinates by carries from the block's first tile (one 64-bit division per thread, not per store).
  const int HtWt = Ht * Wt;
  const int bimg0 = (int)(t0 / HtWt);
  const int rem0 = (int)(t0 - (long long)bimg0 * HtWt);
  const int ty0 = rem0 / Wt, tx0 = rem0 - ty0 * Wt;
  const int re = (tid >> 4) & 3, rc4 = (tid & 15) * 4;
  int s_img[8], s_tyx[8];   // image; ty | tx << 16 (ty = 0xffff: not this block's tile)
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int tl = wave + NW * r;
    const unsigned n = (unsigned)(tx0 + tl), q = n / (unsigned)Wt, tx = n - q * (unsigned)Wt;
    const unsigned m = (unsigned)ty0 + q, q2 = m / (unsigned)Ht, ty = m - q2 * (unsigned)Ht;
    s_img[r] = bimg0 + (int)q2;
    s_tyx[r] = t0 + tl < tend ? (int)(ty | (tx << 16)) : 0xffff;
  }
  if (POOL != 1) {
    // four passes, one output row a of the 4x4 tiles each: [tile][e][channel]
#pragma unroll
    for (int a = 0; a < 4; a++) {
      float* sY = smem + (a & 1) * SYF;
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int tl = 32 * wm + 16 * b + 4 * lk + i;
#pragma unroll
          for (int e = 0; e < 4; e++) sY[(tl * 4 + e) * 64 + wcol] = YO(b, i, 4 * a + e);
        }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int tl = wave + NW * r;
        const int ty = s_tyx[r] & 0xffff, tx = s_tyx[r] >> 16;
        const int oy = 4 * ty + a, ox = 4 * tx + re;
        if (oy < H && ox < W) {   // ty = 0xffff fails here
          const v4f o4 = *reinterpret_cast<const v4f*>(&sY[(tl * 4 + re) * 64 + (rc4 ^ (16 * ((tl >> 2) & 3)))]);
          if constexpr (ABL & 128) { if (o4[0] == 12345.678f) y[tid] = o4[1]; }   // (ablation: the staged row is read, not stored)
          else if constexpr (ABL & 512) __builtin_nontemporal_store(o4, reinterpret_cast<v4f*>(y + (((long long)s_img[r] * H + oy) * W + ox) * Cout + cb * WM_BC + rc4));
          else *reinterpret_cast<v4f*>(y + (((long long)s_img[r] * H + oy) * W + ox) * Cout + cb * WM_BC + rc4) = o4;
        }
      }
    }
  }
  if (POOL != 0) {
    // pooled 2x2 windows of the 4x4 tile: [tile][2 a' + e'][channel]
    float* yp = POOL == 1 ? y : ypool;
    float* sY = smem;   // pass "4": buffer 0 again (its readers of pass 2 are behind barrier 3)
    const int Hp = H / 2, Wp = W / 2;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int tl = 32 * wm + 16 * b + 4 * lk + i;
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
          for (int e2 = 0; e2 < 2; e2++) {
            float p = YO(b, i, 4 * (2 * a2) + 2 * e2);
            const float p1 = YO(b, i, 4 * (2 * a2) + 2 * e2 + 1), p2 = YO(b, i, 4 * (2 * a2 + 1) + 2 * e2),
                        p3 = YO(b, i, 4 * (2 * a2 + 1) + 2 * e2 + 1);
            p = p1 > p ? p1 : p;
            p = p2 > p ? p2 : p;
            p = p3 > p ? p3 : p;
            sY[(tl * 4 + 2 * a2 + e2) * 64 + wcol] = p;
          }
      }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int tl = wave + NW * r;
      const int ty = s_tyx[r] & 0xffff, tx = s_tyx[r] >> 16;
      const int py = 2 * ty + (re >> 1), px = 2 * tx + (re & 1);
      if (py < Hp && px < Wp)
        *reinterpret_cast<v4f*>(yp + (((long long)s_img[r] * Hp + py) * Wp + px) * Cout + cb * WM_BC + rc4) =
            *reinterpret_cast<const v4f*>(&sY[(tl * 4 + re) * 64 + (rc4 ^ (16 * ((tl >> 2) & 3)))]);
    }
  }
}

#undef YO

// Split-Cin reduction: y = [ReLU](sum_s part[s] + bias) [2x2 max-pooled], partials in ascending order.
// POOL as in the main kernel. One thread per float4 of the (pooled, for POOL != 0) output.
template <int POOL>
__global__ __launch_bounds__(256) void wino43_splitk_reduce_kernel(
    const float* __restrict__ part, int S, long long stride, const float* __restrict__ bias, int relu,
    float* __restrict__ y, float* __restrict__ ypool, int B, int H, int W, int C, int imgs_per_group)
{
  const int c4n = C >> 2;
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const long long total = (long long)B * Ho * Wo * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n) * 4;
    long long r = i / c4n;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const v4f bv = *reinterpret_cast<const v4f*>(bias + (size_t)(b / imgs_per_group) * C + c4);
    v4f best = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < (POOL ? 4 : 1); q++) {
      const int yy = POOL ? 2 * oy + (q >> 1) : oy, xx = POOL ? 2 * ox + (q & 1) : ox;
      const long long o = (((long long)b * H + yy) * W + xx) * C + c4;
      v4f v = *reinterpret_cast<const v4f*>(part + o);
      for (int s2 = 1; s2 < S; s2++) v += *reinterpret_cast<const v4f*>(part + s2 * stride + o);
      v += bv;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (POOL != 1) *reinterpret_cast<v4f*>(y + o) = v;
      if (q == 0) best = v;
      else {
#pragma unroll
        for (int e = 0; e < 4; e++) best[e] = v[e] > best[e] ? v[e] : best[e];
      }
    }
    if (POOL != 0) {
      float* yp = POOL == 1 ? y : ypool;
      *reinterpret_cast<v4f*>(yp + (((long long)b * Ho + oy) * Wo + ox) * C + c4) = best;
    }
  }
}

// Cin split of a launch too small to fill the chip (batch-1 conv4_x / conv5_x: 80 / 24 workgroups of 288
// stages each for 512 slots — pure per-workgroup latency): double while the grid stays within the slots
int wino43_cin_split(long long nbt, int ncb, int Cin)
{
  const int NK = Cin / WM_KC;
  int S = 1;
  // Measured round 3 on one frame (tools/bench_wino_mfma.py --batch 1 --groups 1, conv4_2: 80 workgroups of 288 stages):
  // S = 1 186 us, 2 111, 4 107, 8 129 (the reduction reads S partial outputs); conv3_x (152 workgroups) 104 us split or
  // not. A 6-deep LDS ring (prefetch distance 5, one workgroup per CU) made every small launch SLOWER (conv4_2 107 -> 126,
  // conv5_x 66 -> 82), and with one workgroup per CU ring depths 3 / 4 / 5 time the same (conv4_2, S = 1: 184 / 186 / 188 us;
  // S = 2: 111 / 114 / 113): a lone workgroup keeps its MFMA pipe ~65 % busy whatever the prefetch depth — what a small
  // launch lacks is a second workgroup per CU, not operand cover.
  // Round 4: 16-tile workgroups for these launches (the kernel above with one A block per wave: twice the workgroups, half
  // the matrix work each, 176 VGPRs; bit-identical) — conv3_2 at one frame 102 -> 97 us, conv4_x 103 -> 103, conv5_x 42 -> 44,
  // the one-frame trunk 0.893 -> 0.888 ms: a half-size workgroup waits as long for its operands as a full one. Dropped.
  if (nbt * ncb >= 128) return 1;   // (152 workgroups — batch-1 conv3_x — measured slower split in two: the partials cost more than they buy)
  while (S < 8 && nbt * ncb * S * 2 <= 512 && NK % (2 * S) == 0) S *= 2;
  return S;
}

}  // namespace

extern "C" int pcnn_winograd43_conv_workspace_bytes(int B, int H, int W, int Cin, int Cout, int groups, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "winograd43_conv_workspace_bytes: NULL output");
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 64 && Cout >= 64 && groups >= 1 && B % groups == 0, PCNN_EINVAL,
               "winograd43_conv_workspace_bytes: bad shape");
  const long long tpg = (long long)B * ((H + 3) / 4) * ((W + 3) / 4) / groups;
  const int S = wino43_cin_split((long long)groups * ((tpg + 31) / 32), Cout / WM_BC, Cin);
  *bytes = S > 1 ? sizeof(float) * ((size_t)S * B * H * W * Cout + (size_t)groups * Cout) : 0;
  return PCNN_OK;
}

extern "C" int pcnn_winograd43_conv_fwd(const float* v, const float* ut, const float* bias, int B, int H,
                                        int W, int Cin, int Cout, int groups, int relu, int pool,
                                        float* y, float* y_pool, void* workspace, size_t workspace_bytes,
                                        void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1, PCNN_EINVAL, "winograd43_conv: bad shape %dx%dx%d", B, H, W);
  PCNN_REQUIRE(Cin >= 64 && Cin % 64 == 0, PCNN_EINVAL, "winograd43_conv: input channels must be a multiple of 64 (got %d)", Cin);
  PCNN_REQUIRE(Cout >= 64 && Cout % 64 == 0, PCNN_EINVAL, "winograd43_conv: output channels must be a multiple of 64 (got %d)", Cout);
  PCNN_REQUIRE(pool >= 0 && pool <= 2, PCNN_EINVAL, "winograd43_conv: pool must be 0 (none), 1 (pooled only) or 2 (both)");
  PCNN_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), PCNN_EINVAL, "winograd43_conv: pooling needs even height/width");
  PCNN_REQUIRE(groups >= 1 && B % groups == 0, PCNN_EINVAL, "winograd43_conv: batch %d is not a multiple of groups %d", B, groups);
  PCNN_REQUIRE(v && ut && bias && y && (pool != 2 || y_pool), PCNN_ENULL, "winograd43_conv: NULL pointer");
  PCNN_REQUIRE(aligned16(v) && aligned16(ut) && aligned16(bias) && aligned16(y) && (pool != 2 || aligned16(y_pool)), PCNN_EINVAL,
               "winograd43_conv: pointers (v, ut, bias, y, y_pool) must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const long long T = (long long)B * Ht * Wt;
  const long long tpg = T / groups;
  const int ncb = Cout / WM_BC;
  // 32-tile blocks, two independent 4-wave workgroups per CU, for every layer: the pair drifts out of phase,
  // so one workgroup's barrier waits, column folds, epilogue stores and block turnover run under the other's
  // MFMAs. Measured against 64-tile blocks (one 8-wave workgroup per CU, 2/3 of the operand bytes per flop,
  // still instantiated by tools/wino_ablate.hip): conv3_2 1.87 -> 1.75 ms, conv3_1 1.02 -> 0.95, conv4_2
  // 1.79 -> 1.78, 12-layer total 16.25 -> 15.86 ms.
  const long long nbt = (long long)groups * ((tpg + 31) / 32);
  const long long blocks = ((nbt + 7) / 8) * 8 * ncb;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "winograd43_conv: grid too large");
  // the kernel addresses a plane of V and of U^T with 32-bit byte offsets from 64-bit plane bases
  PCNN_REQUIRE(T * Cin < (1ll << 30) && (long long)Cout * Cin < (1ll << 30), PCNN_EINVAL,
               "winograd43_conv: a transform plane of %lld x %d (or %d x %d) floats exceeds the kernel's 32-bit byte offsets", T, Cin, Cout, Cin);
  // optional Cin split for launches that cannot fill the chip (needs the caller's workspace)
  int S = wino43_cin_split(nbt, ncb, Cin);
  const size_t out_elems = (size_t)B * H * W * Cout;
  if (S > 1 && !(workspace && aligned16(workspace) && workspace_bytes >= sizeof(float) * (S * out_elems + (size_t)groups * Cout))) S = 1;
  // Block map (see the kernel). cb-major: every XCD owns one channel block and streams its eighth of the filter bank
  // once; pays when U outweighs V — the deep layers of a single frame (conv4_2: U 37.7 MB, V 22 MB: fabric traffic
  // 8 U + V = 324 MB tile-block-major, U + 8 V = 215 MB channel-block-major; conv5_x 307 -> 85 MB: 64 -> 42 us).
  // PCNN_WINO_MODE overrides the map for experiments and the variant-equality test: 1 = cb-major wherever ncb == 8,
  // 0 = never; unset = the library's choice. (The round-4 kernel variants this switch also selected — the one-wave-per-
  // SIMD 32x32x2 kernel, the round-3 zeroing v_movs — left the library in round 5: tools/variants/)
  static const int env_mode = [] { const char* e = getenv("PCNN_WINO_MODE"); return e ? atoi(e) : -1; }();
  const double u_bytes = 36.0 * Cout * (double)Cin * 4.0 * groups, v_bytes = 36.0 * (double)T * Cin * 4.0;
  int mode = (ncb == 8 && u_bytes > v_bytes) ? 1 : 0;
  if (env_mode >= 0) mode = ((env_mode & 1) && ncb == 8) ? 1 : 0;
  const long long nblocks = (mode & 1) ? 8 * nbt : blocks;
#define WM_GO(P, Y, BIAS, RELU, KS, STRIDE) PCNN_LAUNCH((wino43_mfma_kernel<P, 1, 0, 1>), dim3((unsigned)nblocks, KS), dim3(256), 0, stream, v, ut, BIAS, Y, y_pool, \
                             H, W, Cin, Cout, Ht, Wt, T, tpg, RELU, (int)nbt, ncb, KS, (long long)(STRIDE), mode)
  if (S == 1) {
    if (pool == 0) WM_GO(0, y, bias, relu, 1, 0); else if (pool == 1) WM_GO(1, y, bias, relu, 1, 0); else WM_GO(2, y, bias, relu, 1, 0);
  } else {
    float* part = static_cast<float*>(workspace);
    WM_GO(0, part, (const float*)nullptr, 0, S, out_elems);
    const long long items = (long long)B * (pool ? (H / 2) * (W / 2) : H * W) * (Cout / 4);
    const unsigned rgrid = (unsigned)((items + 255) / 256 < 65536 ? (items + 255) / 256 : 65536);
    if (pool == 0)
      PCNN_LAUNCH(wino43_splitk_reduce_kernel<0>, dim3(rgrid), dim3(256), 0, stream, part, S, (long long)out_elems, bias, relu, y, y_pool, B, H, W, Cout, B / groups);
    else if (pool == 1)
      PCNN_LAUNCH(wino43_splitk_reduce_kernel<1>, dim3(rgrid), dim3(256), 0, stream, part, S, (long long)out_elems, bias, relu, y, y_pool, B, H, W, Cout, B / groups);
    else
      PCNN_LAUNCH(wino43_splitk_reduce_kernel<2>, dim3(rgrid), dim3(256), 0, stream, part, S, (long long)out_elems, bias, relu, y, y_pool, B, H, W, Cout, B / groups);
  }
  return check_launch("winograd43_conv_fwd");
}
