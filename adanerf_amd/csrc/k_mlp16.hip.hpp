// A5 + A6: 16-bit MFMA engine (LDS weight ring, layer_16), the fused PE + shading MLP kernels shade_mlp16_kernel /
// shade_mlp32_kernel and the debug kernel shade_features_kernel.
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_common.hip.hpp"
#include "k_mlp_f32.hip.hpp"
#include "tuning.hpp"
#include <utility>

namespace adanerf {

// ------------------------------------------------------------------------------------------
// A5 + A6: fused PE + shading MLP
// ------------------------------------------------------------------------------------------

struct ShadeArgs {
  ShadeParams sp;
  NetParams net;
  const float* rays;          // [*,8]
  const uint32_t* sample_key; // [S]; null in dense mode (128 samples per ray in order: the key of sample i is i)
  const float* sample_z;      // [S] world depth per sample (inverse-CDF sampler); null -> ztab[bin]
  const int32_t* total;       // device S (may be null -> max_samples)
  int32_t max_samples;
  float* raw_out;             // [S,4]
};

// one sample's raw outputs, back in the network's own scale (NetParams::out_scale: 1, or the exact power of two the bf16 packing took out)
__device__ __forceinline__ void store_raw(const ShadeArgs& a, int s, float r, float g, float b, float alpha) {
  const float ks = a.net.out_scale[1];
  *reinterpret_cast<float4*>(a.raw_out + static_cast<size_t>(s) * 4) = make_float4(r * ks, g * ks, b * ks, alpha * a.net.out_scale[0]);
}

// kClampRelu: ReLU + conversion of a pair is ONE instruction, `v_cvt_pk_bf16_f32 ... clamp` (clamps to [0, 1]; gfx950 honours the modifier on
// this conversion: tools/probes/cvt_clamp_probe.hip), instead of v_cvt_pk + v_pk_max_i16.  Legal because the bf16 packing scales every layer by
// a power of two chosen from a rigorous bound on its activations, so that nothing a ReLU layer can produce exceeds 1 (pack.cpp scale_layer);
// powers of two commute with every rounding on the way, so the un-scaled outputs are those of the unscaled network bit for bit.  fp16 has no
// range for this.  Measured: 8 x 256 shading kernel -1.8 %, 5 x 256 -5.7 %, 6 x 128 -2 %, 4 x 64 -3.3 % (profiles/r05_lab_log.md 1).
struct Bf16 {
  static constexpr bool kClampRelu = true;
  typedef bf16x8 vec8;
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct Fp16 {
  static constexpr bool kClampRelu = false;
  typedef f16x8 vec8;
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// ---- weight streaming through LDS -------------------------------------------------------------
// The packed 16-bit shading net is one linear stream of 1 KiB A fragments in consumption order
// (layer, tile m, k-step s).  All waves of a workgroup consume it in lockstep, so it is staged ONCE
// per workgroup: 8 KiB chunks (8 fragments; each of the 8 waves DMA-copies one fragment with
// global_load_lds_dwordx4, LDS image lane-linear = fragment order, so ds_read_b128 is conflict
// free) into a 4-slot ring.  One s_barrier per chunk; counted vmcnt keeps 2-3 chunks in flight
// across the barrier (never vmcnt(0) in the loop).  Each wave keeps the current chunk's 8
// fragments in registers and re-fills fragment i from the NEXT chunk right after the MFMA that
// consumed it, so LDS latency hides behind the other 7 MFMAs.
// Ring geometry, register ring depth, stagger / DMA grouping and the timing-ablation bits: tuning.hpp.
using tune::kRegFrags;
constexpr int kShadeFrags16 = 32 + 4 * 128 + 160 + 2 * 128 + 144 + 72 + 8;   // 1184 per pass (FP=10, FD=4)
constexpr int kShadeBiasFloats = 8 * 256 + 288 + 128 + 32;                     // 2496

// CF / RS / LPW (fragments each wave DMA-copies per chunk = CF / waves) are compile-time; the slot a
// chunk lives in is a run-time counter, so any tile length that is a multiple of CF works.
template <int CF, int RS, int LPW, int NR = kRegFrags>
struct WStream {
  static constexpr int kRegs = NR;   // fragments held in registers = re-fill distance
  static constexpr int kChunk = CF;  // fragments per chunk
  static constexpr int kChunkBytes = CF * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc;   // raw buffer descriptor of the stream
#endif
  const char* gbase;     // stream start (global)
  uint32_t gbytes;       // stream length in bytes (multiple of kChunkBytes)
  uint32_t goff;         // byte offset of the next chunk to issue
  uint32_t lane_off;     // lane * 16
  uint32_t wave_off;     // byte offset of this wave's first fragment inside a chunk
  uint32_t slot_cur;     // ring slot of the chunk being consumed (wave-uniform)
  uint32_t rd_cur;       // LDS byte address of (current chunk, this lane)
  uint32_t rd_next;      // LDS byte address of (next chunk, this lane)
  uint32_t lds_base;     // LDS byte address of the ring
  uint32_t grp;          // 0: this wave synchronises at chunk position 0, 1: at position CF / 2 (see ws_sync)
  uint32_t issuer;       // this wave issues LDS-DMA pieces (all waves, or one group only: tune::kDmaGroup)
  uint32_t dma_lds;      // the chunk being copied: LDS byte address of this wave's first piece (M0 of its LPW copies) ...
  uint32_t dma_goff;     // ... and the stream byte offset of that piece (scalar offset of its LPW copies); set at the wave's synchronisation point
  bool stag;             // the workgroup runs its two wave groups half a chunk apart (compile-time constant per kernel)
  u32x4 R[NR];           // register ring: fragment p (position inside the chunk) lives in R[p % NR]
};

__device__ __forceinline__ u32x4 lds_read128(uint32_t byte_addr) {
  typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
  return *((lds_u32x4_ptr)(uintptr_t)byte_addr);
}

// One LDS-DMA piece (1 KiB: lane l copies 16 bytes) of the chunk st.dma_lds / st.dma_goff describe.  The instruction's immediate offset
// moves BOTH the global address and the LDS destination (tools/probes/lds_dma_offset.hip, profiles/r06_lds_dma_offset.log), so the LPW
// pieces of a chunk share one M0 value and one scalar offset: no per-piece scalar arithmetic, no per-piece M0 write.  Round 5 re-computed
// both per piece and issued the LPW pieces back to back behind the barrier: 5.0 cycles per MFMA of a wave that is alone on its SIMD;
// one M0 + immediates: 2.0; the same pieces spread over the chunk, one every CF / LPW fragments: 0.85 (tools/probes/dma_issue_cost.hip,
// profiles/r06_dma_issue_cost.log).
template <int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_piece(WStream<CF, RS, LPW, NR>& st, int i) {      // i: compile-time after unrolling
  static_assert(LPW >= 1 && LPW <= 8, "a piece is addressed by the 12-bit immediate offset of buffer_load ... lds (pieces 4 .. 7: a second base)");
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const uint32_t far = i >= 4 ? 4096u : 0u;
  lds_ptr dst = (lds_ptr) static_cast<uintptr_t>(st.dma_lds + far);
  const int voff = static_cast<int>(st.lane_off), soff = static_cast<int>(st.dma_goff + far);
  if (tune::kAblateDmaBytes) {      // timing ablation (wrong results): the same instructions moving a quarter of the bytes (4 per lane)
    if ((i & 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 4, voff, soff, 0, 0);
    else if ((i & 3) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 4, voff, soff, 1024, 0);
    else if ((i & 3) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 4, voff, soff, 2048, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 4, voff, soff, 3072, 0);
    return;
  }
  // buffer form (buffer_load_dwordx4 ... lds): descriptor + wave-uniform byte offset in SGPRs, lane * 16 as the only VGPR operand
  if ((i & 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 16, voff, soff, 0, 0);
  else if ((i & 3) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 16, voff, soff, 1024, 0);
  else if ((i & 3) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 16, voff, soff, 2048, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, dst, 16, voff, soff, 3072, 0);
#endif
}

// the next chunk of the stream goes to ring slot `slot`: describe it (ws_piece copies it) and advance the stream
template <int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_chunk_begin(WStream<CF, RS, LPW, NR>& st, uint32_t slot) {
  st.dma_lds = st.lds_base + slot * (CF * 1024) + st.wave_off;
  st.dma_goff = st.goff + st.wave_off;
  st.goff += CF * 1024;
  if (st.goff >= st.gbytes) st.goff = 0;
}

// a whole chunk at once (kernel prologue)
template <int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_issue(WStream<CF, RS, LPW, NR>& st, uint32_t slot) {
  ws_chunk_begin(st, slot);
#pragma unroll
  for (int i = 0; i < LPW; ++i) ws_piece(st, i);
}

// Synchronisation point k of the ring: own pieces of chunk k+1 have landed (<= (RS-3) LPW younger DMAs outstanding);
// barrier => chunk k+1 is complete in LDS for every wave and every wave has consumed chunk k-1 (its MFMAs were issued
// before the barrier, so its ds_reads returned) => refill the slot of chunk k-1 with chunk k+RS-1.
// Waves of group 0 reach it at fragment position 0 of chunk k, waves of group 1 at position CF / 2 of chunk k (tune::kStagger):
// the two waves that share a SIMD (wave i and wave i + 4 of an 8-wave workgroup) then run half an output tile apart, so
// one of them is issuing MFMAs while the other is in its bias-read / epilogue / DMA-issue phase, instead of both hitting
// those phases in the same cycles (a workgroup barrier per chunk otherwise keeps all eight waves in lockstep).  Both
// groups see the same guarantees: a wave of group 1 is at most half a chunk ahead, i.e. still inside chunk k.
// see ws_position
__device__ __forceinline__ void ws_skip_pad() {
  if (tune::kSkipPad >= 0) asm volatile("s_nop %0" ::"n"(tune::kSkipPad < 0 ? 0 : tune::kSkipPad) : "memory");
}

template <int ABL, int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_sync(WStream<CF, RS, LPW, NR>& st, bool pad, int piece) {
  static_assert(RS >= 3 && (RS - 2) * LPW < 64, "vmcnt is a 6-bit counter");
  if (ABL & 1) return;
  if (!(ABL & 32)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RS - 3) * LPW) : "memory");   // 32: no wait/barrier
  // slot of chunk k-1: group 0 is still "in" it (ws_advance follows), group 1 has moved on to chunk k
  const uint32_t slot = (st.grp == 0) ? st.slot_cur : (st.slot_cur == 0 ? RS - 1 : st.slot_cur - 1);
  if (!(ABL & 16)) {                                                                                                      // 16: no DMA
    if (st.issuer) {
      ws_chunk_begin(st, slot);
      ws_piece(st, piece);      // the other pieces of the chunk follow at their positions (ws_position)
    } else if (pad) ws_skip_pad();      // a barrier that releases at once is not 11 wait states
  }
}

// the wave moves from chunk k-1 to chunk k: rotate the LDS read addresses
template <int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_advance(WStream<CF, RS, LPW, NR>& st) {
  st.slot_cur = (st.slot_cur + 1 == RS) ? 0 : st.slot_cur + 1;
  const uint32_t nxt = (st.slot_cur + 1 == RS) ? 0 : st.slot_cur + 1;
  st.rd_cur = st.rd_next;
  st.rd_next = st.lds_base + nxt * (CF * 1024) + st.lane_off;
}

// fragment position f (compile-time) of the stream is about to be consumed
// The group that has no synchronisation point at this position jumps over it -- the only branches inside the MFMA stream.
// ws_skip_pad(): the MFMA -> VALU read-after-write distance (11 wait states for these 8-pass MFMAs) is inserted by the
// compiler, and on the short side of such a branch it came out too small (7 wait states between the last MFMA of a
// tile and the first v_cvt_pk of its epilogue, hipcc / ROCm 7.2 with -amdgpu-mfma-vgpr-form: the epilogue then read
// accumulator rows the last MFMA had not written yet -- wrong rgb for waves 4-7 in the variants of
// profiles/r02_stagger_hazard.md).  s_nop kSkipPad = 12 explicit wait states on the short side (11 are required) make the
// distance independent of what the scheduler puts after the join.

// tile_start: the instruction stream in front of this position ends with the last MFMA of an output tile (whose epilogue
// the compiler may have moved behind the branch); mid-tile, the next consumer of the accumulator is the next MFMA of the chain.
template <int ABL, int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_position(WStream<CF, RS, LPW, NR>& st, int f, bool tile_start) {
  // The LPW pieces of a chunk are issued CF / LPW fragment positions apart: piece f / STEP at position f, whichever group the wave is in --
  // a wave of group 1 starts its chunk at position CF / 2 with piece LPW / 2 and wraps (pieces are independent quarters of the chunk).
  // RS == 3 leaves a chunk one chunk time to land: its pieces go into the first half of the chunk (STEP halved); such a ring is not staggered.
  constexpr int STEP = RS == 3 ? CF / LPW / 2 : CF / LPW;
  static_assert(CF % LPW == 0 && STEP >= 1 && (RS == 3 || (CF / 2) % STEP == 0), "piece positions: both groups' synchronisation points are piece positions");
  const bool pad = tile_start;
  const bool piece_pos = f % STEP == 0 && f / STEP < LPW && !(ABL & (1 | 16));
  if (f == 0) {
    if (ABL & 1) return;
    if (st.grp == 0) ws_sync<ABL>(st, pad, 0);
    else {
      if (pad) ws_skip_pad();
      if (piece_pos && st.issuer) ws_piece(st, 0);
    }
    ws_advance(st);
  } else if (f == CF / 2 && st.stag) {
    if (st.grp != 0) ws_sync<ABL>(st, pad, (CF / 2) / STEP);
    else {
      if (pad) ws_skip_pad();
      if (piece_pos && st.issuer) ws_piece(st, (CF / 2) / STEP);
    }
  } else if (piece_pos) {
    if (st.issuer) ws_piece(st, f / STEP);
  }
}

// fragment position p inside the current chunk has just been consumed: re-fill its register with fragment
// p + kRegFrags (same chunk, or the next chunk -- already landed: see ws_sync)
template <int ABL, int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_refill(WStream<CF, RS, LPW, NR>& st, int p) {
  if (ABL & 2) {
    asm volatile("" : "+v"(st.R[p % NR]));
    return;
  }
  const int q = p + NR;
  st.R[p % NR] = (q < CF) ? lds_read128(st.rd_cur + q * 1024) : lds_read128(st.rd_next + (q - CF) * 1024);
}

template <int CF, int RS, int LPW, int NR>
__device__ __forceinline__ void ws_start(WStream<CF, RS, LPW, NR>& st, const void* gbase, uint32_t gbytes, char* lds, int wave, int lane,
                                         int grp = -1, bool issuer = true) {
  // wave: which pieces of a chunk this wave DMA-copies; grp: -1 = the workgroup is not staggered, else the wave's group (0 / 1)
  st.stag = grp >= 0;
  st.grp = grp > 0 ? 1u : 0u;
  st.issuer = issuer;
  st.gbase = reinterpret_cast<const char*>(gbase);
  st.gbytes = gbytes;
#if defined(__HIP_DEVICE_COMPILE__)
  st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gbase), 0, static_cast<int>(gbytes), 0x00020000);
#endif
  st.goff = 0;
  st.lane_off = lane * 16;
  st.wave_off = wave * LPW * 1024;
  st.lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
  if (issuer) {
#pragma unroll
    for (int k = 0; k < RS - 1; ++k) ws_issue(st, k);
  }
  // st.dma_* now describe chunk RS - 2.  A staggered group-1 wave passes piece positions before its first synchronisation point: those
  // copies repeat pieces of chunk RS - 2 (same bytes to the same place, long before anyone reads them); being extra YOUNGER copies they
  // only make the first counted waits stricter.
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RS - 2) * LPW) : "memory");
  st.slot_cur = RS - 1;                       // the first ws_advance moves to slot 0 = chunk 0
  st.rd_cur = st.lds_base + st.lane_off;      // unused until then
  st.rd_next = st.lds_base + st.lane_off;     // chunk 0
#pragma unroll
  for (int i = 0; i < NR; ++i) st.R[i] = lds_read128(st.rd_next + i * 1024);
}

// ReLU on the raw bits: max(int(x), 0) is +0.0 for every negative float and the identity for positive
// ones -- one v_max_i32, no canonicalising v_max_f32 pair.
__device__ __forceinline__ float relu_bits(float x) {
  int i = __builtin_bit_cast(int, x);
  i = i > 0 ? i : 0;
  return __builtin_bit_cast(float, i);
}

template <class ET, int F>
__device__ __forceinline__ void pe_pack(const float x[3], int h, uint32_t* out) {
  float t[pe_slots(F)];
  pe_eval<F, false>(x, h, t);
#pragma unroll
  for (int q = 0; q < pe_slots(F) / 2; ++q) out[q] = ET::pack(t[2 * q], t[2 * q + 1]);
}

// Bias block [m][h][16] for this lane-half from LDS with hand-issued reads: hipcc cannot see an asm
// ds_read, so it neither assumes aliasing with the LDS-DMA ring (which costs an s_waitcnt vmcnt(0) drain
// per tile) nor needs the 3-VALU-per-value SGPR select that scalar loads cost.  The wait statement names
// every destination "+v" so no consumer is scheduled above it (cdna_hip_programming.md 5.7 form ii).
struct BiasRegs {
  f32x4 b0, b1, b2, b3;
};
// issue now, consume later: the reads stay in flight behind the MFMAs of the current tile
__device__ __forceinline__ void lds_bias_issue(uint32_t byte_addr, BiasRegs& r) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
               : "=&v"(r.b0), "=&v"(r.b1), "=&v"(r.b2), "=&v"(r.b3)
               : "v"(byte_addr));
}
// The same request with the two LDS addresses every fragment re-fill of the wave is computed from passed through the statement ("+v"): a
// re-fill issued behind it in program order cannot be scheduled in front of it, which is what a COUNTED wait for the block relies on
// (lds_bias_take<YOUNGER>: the re-fills of the tile are younger than the request).
__device__ __forceinline__ void lds_bias_issue(uint32_t byte_addr, BiasRegs& r, uint32_t& rd0, uint32_t& rd1) {
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\tds_read_b128 %3, %6 offset:48"
               : "=&v"(r.b0), "=&v"(r.b1), "=&v"(r.b2), "=&v"(r.b3), "+v"(rd0), "+v"(rd1)
               : "v"(byte_addr));
}
// YOUNGER: LDS reads this wave has issued behind the bias reads that need not have returned (LDS returns in order; the
// counter has 4 bits).  0 drains every fragment prefetch in flight -- what a wave alone on its SIMD should not do per tile.
template <int YOUNGER = 0>
__device__ __forceinline__ void lds_bias_take(BiasRegs& r, f32x16* acc) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r.b0), "+v"(r.b1), "+v"(r.b2), "+v"(r.b3) : "n"(YOUNGER > 15 ? 15 : YOUNGER));
  f32x16 a;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = r.b0[e];
    a[4 + e] = r.b1[e];
    a[8 + e] = r.b2[e];
    a[12 + e] = r.b3[e];
  }
  *acc = a;
}

__device__ __forceinline__ void lds_bias16(uint32_t byte_addr, f32x16* acc) {
  f32x4 b0, b1, b2, b3;
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
               : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
               : "v"(byte_addr));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  f32x16 a;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = b0[e];
    a[4 + e] = b1[e];
    a[8 + e] = b2[e];
    a[12 + e] = b3[e];
  }
  *acc = a;
}

// One 16-bit layer for one 32-sample column block.  Input = two register segments (S1 then S2
// k-steps of 8 slots = 4 packed dwords each); output tile m lands in out[8m .. 8m+7] (packed pairs).
// FPOS = position of the layer's first fragment in the stream modulo the chunk size.
// KEEP_F32_TILE >= 0: that tile's raw accumulator is returned in *keep instead (alpha / rgb rows);
// kKeepAllF32: all of them, in keep[0 .. MT-1] (the sampling net's 128 raw outputs).
// epilogue of one accumulator quad g (values 4g..4g+3 of tile m): convert, ReLU on the packed pairs
// The clamped conversion has no compiler-visible form (hipcc folds a clamp into VOP3 arithmetic, not into this conversion), so it is inline
// asm -- and the hazard recogniser does not look inside inline asm: it inserts NO wait states between an MFMA and an asm statement that reads
// the MFMA's result (checked on the ISA: `s_nop 11` in front of a plain v_cvt_pk, nothing in front of the asm; the first version of this read
// accumulator rows the last MFMA had not written yet).  mfma_guard: one compiler-visible VALU read of the finished accumulator
// (v_readfirstlane: the recogniser pads THAT as the MFMA -> VALU rule requires); every asm conversion of the tile takes its result as an
// operand, so none can be scheduled in front of it.  One guard per 16 values.
template <class ET, bool RELU>
__device__ __forceinline__ int mfma_guard(const f32x16& acc) {
  if constexpr (RELU && ET::kClampRelu) return __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[15]));
  else return 0;
}

template <class ET, bool RELU>
__device__ __forceinline__ void epilogue_quad_16(const f32x16& acc, int m, int g, uint32_t* out, int guard = 0) {
  if constexpr (RELU && ET::kClampRelu) {
    uint32_t q0, q1;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2 clamp\n\t; after the accumulator's guard %3" : "=v"(q0) : "v"(acc[4 * g + 0]), "v"(acc[4 * g + 1]), "s"(guard));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2 clamp\n\t; after the accumulator's guard %3" : "=v"(q1) : "v"(acc[4 * g + 2]), "v"(acc[4 * g + 3]), "s"(guard));
    out[8 * m + 2 * g + 0] = q0;
    out[8 * m + 2 * g + 1] = q1;
    return;
  }
  // convert first, then ReLU on the packed pair: max(int16, 0) clears every negative bf16/f16
  // (one v_cvt_pk + one v_pk_max_i16 per two values)
  uint32_t p0 = ET::pack(acc[4 * g + 0], acc[4 * g + 1]), p1 = ET::pack(acc[4 * g + 2], acc[4 * g + 3]);
  if (RELU) {
    const s16x2 z = {0, 0};
    p0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p0), z));
    p1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p1), z));
  }
  out[8 * m + 2 * g + 0] = p0;
  out[8 * m + 2 * g + 1] = p1;
}

constexpr int kKeepAllF32 = -2;   // layer_16 KEEP_F32_TILE: every tile's raw accumulator goes to keep[m]


template <class ET, class WS, int S1, int S2, int MT, bool RELU, int FPOS, int KEEP_F32_TILE = -1>
__device__ __forceinline__ void layer_16(WS& st, uint32_t bias_addr, int lane, const uint32_t* in1, const uint32_t* in2,
                                         uint32_t* out, f32x16* keep = nullptr) {
  constexpr int CF = WS::kChunk;
  constexpr int KS = S1 + S2;
  // bias_addr: LDS byte address of this layer's bias block for THIS lane-half ([m][h][16] floats)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc;
    if (tune::kAblateShade & 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    } else {
      lds_bias16(bias_addr + m * 128, &acc);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int f = (FPOS + m * KS + s) % CF;     // position inside the chunk; compile-time after unrolling
      ws_position<tune::kAblateShade>(st, f, s == 0);
      const uint32_t* src = (s < S1) ? (in1 + 4 * s) : (in2 + 4 * (s - S1));
      u32x4 b = {src[0], src[1], src[2], src[3]};
      acc = ET::mfma(st.R[f % WS::kRegs], b, acc);
      ws_refill<tune::kAblateShade>(st, f);
    }
    if (KEEP_F32_TILE == kKeepAllF32) {
      keep[m] = acc;
    } else if (KEEP_F32_TILE == m) {
      *keep = acc;
    } else if (tune::kAblateShade & 8) {
      asm volatile("" ::"v"(acc));
#pragma unroll
      for (int g = 0; g < 8; ++g) asm volatile("" : "=v"(out[8 * m + g]));
    } else {
      const int gd = mfma_guard<ET, RELU>(acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(acc, m, g, out, gd);
    }
  }
}

// Loads the sample's ray record and evaluates position (+ optional unit direction).
__device__ __forceinline__ void load_sample(const ShadeArgs& a, int s, int total, float x[3], float dpe[3]) {
  const int si = (s < total) ? s : (total > 0 ? total - 1 : 0);
  const uint32_t key = a.sample_key ? a.sample_key[si] : static_cast<uint32_t>(si);      // null: dense mode, sample i = (ray i >> 7, bin i & 127)
  const uint32_t ray = key >> 7;
  const int bin = static_cast<int>(key & 127u);
  const float4* rr = reinterpret_cast<const float4*>(a.rays + static_cast<size_t>(ray) * 8);
  const float4 o4 = rr[0], d4 = rr[1];
  const float o[3] = {o4.x, o4.y, o4.z}, d[3] = {d4.x, d4.y, d4.z};
  sample_position(a.sp, o, d, a.sample_z ? a.sample_z[si] : a.sp.ztab[bin], x);
  if (a.sp.unit_dir) unit3(d, dpe);
  else {
    dpe[0] = d[0];
    dpe[1] = d[1];
    dpe[2] = d[2];
  }
}

// Wave-private LDS stash for packed PE slots (hand-issued so hipcc neither orders them against the LDS-DMA
// ring with vmcnt(0) nor keeps 24 VGPRs alive across the layer stack).  Layout [dword group of 4][lane]:
// b128 accesses are lane-linear, hence conflict-free.
template <int NQ>   // NQ = number of b128 groups
__device__ __forceinline__ void lds_stash_write(uint32_t byte_addr, const uint32_t* v) {
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
    const u32x4 t = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr + g * 1024), "v"(t) : "memory");
  }
}
template <int NQ>
__device__ __forceinline__ void lds_stash_read(uint32_t byte_addr, uint32_t* v) {
  u32x4 t[NQ];
#pragma unroll
  for (int g = 0; g < NQ; ++g) asm volatile("ds_read_b128 %0, %1" : "=&v"(t[g]) : "v"(byte_addr + g * 1024));
  if (NQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[NQ > 3 ? 3 : 0]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[NQ > 1 ? 1 : 0]));
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
    v[4 * g] = t[g][0];
    v[4 * g + 1] = t[g][1];
    v[4 * g + 2] = t[g][2];
    v[4 * g + 3] = t[g][3];
  }
}

// A5+A6, 16-bit MFMA path.  Workgroup = WAVES waves x 32 samples; persistent over tiles; the weight
// stream is cyclic so DMA prefetch runs across tile boundaries.  WAVES = 4 with two workgroups per CU
// (two waves per SIMD from DIFFERENT workgroups): each workgroup has its own ring and barriers, so the
// two waves sharing a SIMD are not in lockstep and one computes while the other waits at its barrier.
template <class ET, int FP, int FD, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 2) void shade_mlp16_kernel(ShadeArgs a) {
  static_assert(FP == 10 && FD == 4, "fragment positions below assume the 10-4 shading encoding");
  static_assert(WAVES == 4 || WAVES == 8, "chunk = 8 fragments");
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD);
  constexpr bool kOneGroupDma = tune::kDmaGroup >= 0 && WAVES == 8;
  constexpr int TILE = WAVES * 32, CF = tune::kChunkFrags, RS = tune::kRingSlots, LPW = kOneGroupDma ? CF / 4 : CF / WAVES;
  constexpr int kRingBytes = CF * RS * 1024;
  static_assert(CF % WAVES == 0 && CF % kRegFrags == 0 && kShadeFrags16 % CF == 0 && CF % 8 == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW> WS;
  constexpr int kStashBytes = WAVES * (QP / 8 + QD / 8) * 1024;      // PE slots computed once per tile and parked in LDS
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kShadeBiasFloats * 4 + kStashBytes];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const uint32_t stash = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + kShadeBiasFloats * 4 +
                         wave * (QP / 8 + QD / 8) * 1024 + lane * 16;
  int total = a.total ? *a.total : a.max_samples;
  if (total > a.max_samples) total = a.max_samples;
  const int ntiles = (total + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;    // workgroup-uniform

  {
    float* lds_bias = reinterpret_cast<float*>(lds + kRingBytes);
    for (int i = threadIdx.x; i < kShadeBiasFloats; i += blockDim.x) lds_bias[i] = a.net.bias[i];
  }
  __syncthreads();
  WS st;
  ws_start(st, a.net.w, kShadeFrags16 * 1024, lds, kOneGroupDma ? (wave & 3) : wave, lane,
           (tune::kStagger && WAVES == 8) ? (wave >> 2) : -1, !kOneGroupDma || (wave >> 2) == tune::kDmaGroup);

  // LDS byte address of the bias blocks of this lane-half
  const uint32_t bias0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + h * 64;
  const uint32_t* bo = a.net.b_off;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s = tile * TILE + wave * 32 + j;
    uint32_t hA[64], hB[64];
    {
      float x[3], dpe[3];
      load_sample(a, s, total, x, dpe);
      uint32_t pts[QP / 2];
      pe_pack<ET, FP>(x, h, pts);
      uint32_t dirs[QD / 2];
      pe_pack<ET, FD>(dpe, h, dirs);
      lds_stash_write<QP / 8>(stash, pts);
      lds_stash_write<QD / 8>(stash + (QP / 8) * 1024, dirs);
      layer_16<ET, WS, QP / 8, 0, 8, true, 0>(st, bias0 + bo[0] * 4, lane, pts, pts, hA);
    }
#pragma unroll 1
    for (int l = 1; l <= 3; l += 2) {
      layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[l] * 4, lane, hA, hA, hB);
      layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[l + 1] * 4, lane, hB, hB, hA);
    }
    {
      // the skip connection takes the 32 position slots back from the LDS stash instead of holding 16 VGPRs across layers 1-4
      uint32_t pts[QP / 2];
      lds_stash_read<QP / 8>(stash, pts);
      layer_16<ET, WS, QP / 8, 16, 8, true, 0>(st, bias0 + bo[5] * 4, lane, pts, hA, hB);   // cat([pts, h])
    }
    layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[6] * 4, lane, hB, hB, hA);
    layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[7] * 4, lane, hA, hA, hB);
    f32x16 alpha_tile;
    layer_16<ET, WS, 16, 0, 9, false, 0, 8>(st, bias0 + bo[8] * 4, lane, hB, hB, hA, &alpha_tile);      // feature (+alpha row)
    const float alpha = alpha_tile[0];
    {
      uint32_t dirs[QD / 2];
      lds_stash_read<QD / 8>(stash + (QP / 8) * 1024, dirs);
      layer_16<ET, WS, 16, QD / 8, 4, true, (32 + 4 * 128 + 160 + 2 * 128 + 144) % CF>(st, bias0 + bo[9] * 4, lane, hA, dirs, hB);             // cat([feature, dir])
    }
    f32x16 rgb_tile;
    layer_16<ET, WS, 8, 0, 1, false, (32 + 4 * 128 + 160 + 2 * 128 + 144 + 72) % CF, 0>(st, bias0 + bo[10] * 4, lane, hB, hB, hA, &rgb_tile);
    if (h == 0 && s < total)
      store_raw(a, s, rgb_tile[0], rgb_tile[1], rgb_tile[2], alpha);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// ---- two sample blocks per wave -------------------------------------------------------------------------------------------
// layer_16 for TWO 32-sample column blocks held by one wave: every weight fragment read from LDS feeds two MFMAs (one per
// block), so the LDS operand traffic, the DMA issue and the barriers per MFMA are half of layer_16's.  The price is the register
// file: 2 x (64 + 64) packed activation registers, i.e. one wave per SIMD (up to 512 registers), so nothing but the wave's own
// instruction stream hides latencies: the two accumulator chains alternate (no dependent back-to-back MFMAs), the bias block of
// tile m + 1 is requested while tile m computes, and the epilogue of tile m - 1 (both blocks: 8 quads) is spread over the
// first k-steps of tile m.  Both chains start from the bias registers as the C operand (no copies).
// A finished tile pair (both blocks) whose conversion has not run yet: tune::kShadeCarry hands the LAST tile of a layer to the next layer, which
// converts it under the MFMAs of its own first tile (the outputs feed the last two k-steps of that tile's input segment) instead of at the
// layer boundary with the matrix pipe idle.
struct PendingTile2 {
  f32x16 a, b;
};
// PEND_M >= 0: `pend` holds tile PEND_M of the previous layer (ReLU: PEND_RELU); its packed outputs go to pendA / pendB = the arrays this layer
// reads as (part of) its input.  CARRY_OUT: this layer's last tile is left in `pend` for the next layer.
template <class ET, class WS, int S1, int S2, int MT, bool RELU, int FPOS, int KEEP_F32_TILE = -1, int PEND_M = -1, bool PEND_RELU = true,
          bool CARRY_OUT = false>
__device__ __forceinline__ void layer_16x2(WS& st, uint32_t bias_addr, const uint32_t* in1A, const uint32_t* in2A, const uint32_t* in1B,
                                           const uint32_t* in2B, uint32_t* outA, uint32_t* outB, PendingTile2& pend, uint32_t* pendA = nullptr,
                                           uint32_t* pendB = nullptr, f32x16* keepA = nullptr, f32x16* keepB = nullptr) {
  constexpr int CF = WS::kChunk;
  constexpr int KS = S1 + S2;
  constexpr int ABL = tune::kAblateShade;
  constexpr int PER = (KS >= 8) ? 1 : (8 + KS - 1) / KS;      // epilogue quads of the previous tile per k-step
  constexpr bool kKeepAll = KEEP_F32_TILE == kKeepAllF32;     // every tile's raw accumulators go to keepA[m] / keepB[m] (the sampling net's 128 raw outputs)
  auto kept = [](int m) { return kKeepAll || KEEP_F32_TILE == m; };
  constexpr bool kCarry = tune::kShadeCarry && !(ABL & 8);
  constexpr bool kHasPend = kCarry && PEND_M >= 0;
  constexpr int PERP = 1;                                     // quads of a carried tile per k-step: all converted before the k-steps that read them
  static_assert(!kHasPend || KS >= 16, "a carried tile is consumed by the last two k-steps of a 16-k-step input segment, 6 k-steps behind its last conversion");
  static_assert(!CARRY_OUT || (KEEP_F32_TILE != MT - 1 && KEEP_F32_TILE != kKeepAllF32), "the kept tile is not converted");
  // LDS reads issued between a bias request and its use: the tile's KS fragment re-fills (none under ablation 2)
  constexpr bool kCounted = tune::kBiasWaitCounted && tune::kSchedGroups && !(ABL & (2 | 4));
  constexpr int kYounger = kCounted ? KS : 0;
  BiasRegs br;
  f32x16 pA, pB;
  int gA = 0, gB = 0;      // mfma_guard of the tile whose conversions are pending
  if (kHasPend) {
    pA = pend.a;
    pB = pend.b;
  }
  if (!(ABL & 4)) lds_bias_issue(bias_addr, br, st.rd_cur, st.rd_next);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 bias, accA, accB;
    if (tune::kSchedGroups) __builtin_amdgcn_sched_barrier(0);      // one scheduling region per tile (the group solver's cost grows fast with the region)
    if (ABL & 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = 0.f;
    } else {
      if (m == 0) lds_bias_take<0>(br, &bias);              // first tile of the layer: issued just now
      else lds_bias_take<kYounger>(br, &bias);
      if (m + 1 < MT) lds_bias_issue(bias_addr + (m + 1) * 128, br, st.rd_cur, st.rd_next);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int f = (FPOS + m * KS + s) % CF;
      ws_position<ABL>(st, f, false);
      const uint32_t* sa = (s < S1) ? (in1A + 4 * s) : (in2A + 4 * (s - S1));
      const uint32_t* sb = (s < S1) ? (in1B + 4 * s) : (in2B + 4 * (s - S1));
      const u32x4 ba = {sa[0], sa[1], sa[2], sa[3]}, bb = {sb[0], sb[1], sb[2], sb[3]};
      accA = ET::mfma(st.R[f % WS::kRegs], ba, s == 0 ? bias : accA);
      accB = ET::mfma(st.R[f % WS::kRegs], bb, s == 0 ? bias : accB);
      ws_refill<ABL>(st, f);
      // The previous tile's guards are taken in k-step E0: with a scheduling fence per k-step (below) a guard in k-step 0 sits right behind the tile's
      // first MFMA, one or two MFMAs after the accumulator's last write, and costs an `s_nop 6` per tile (hipcc counts an MFMA as ONE wait state of the 11 it
      // wants between the write and a VALU read); tune::kShadeGuardStep k-steps later the distance is there by itself.
      constexpr int E0 = (tune::kShadeKstepFence && KS >= 8 + tune::kShadeGuardStep) ? tune::kShadeGuardStep : 0;
      if (m > 0 && !kept(m - 1) && !(ABL & 8) && s >= E0) {
        if (s == E0) {
          gA = mfma_guard<ET, RELU>(pA);
          gB = mfma_guard<ET, RELU>(pB);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int q = (s - E0) * PER + k;
          if (q < 4) epilogue_quad_16<ET, RELU>(pA, m - 1, q, outA, gA);
          else if (q < 8) epilogue_quad_16<ET, RELU>(pB, m - 1, q - 4, outB, gB);
        }
      }
      if (m == 0 && kHasPend && s >= E0) {      // the previous layer's last tile
        if (s == E0) {
          gA = mfma_guard<ET, PEND_RELU>(pA);
          gB = mfma_guard<ET, PEND_RELU>(pB);
        }
#pragma unroll
        for (int k = 0; k < PERP; ++k) {
          const int q = (s - E0) * PERP + k;
          if (q < 4) epilogue_quad_16<ET, PEND_RELU>(pA, PEND_M < 0 ? 0 : PEND_M, q, pendA, gA);
          else if (q < 8) epilogue_quad_16<ET, PEND_RELU>(pB, PEND_M < 0 ? 0 : PEND_M, q - 4, pendB, gB);
        }
      }
      if (tune::kSchedGroups) {
        // pin the interleave the source describes: per k-step two MFMAs, the fragment re-fill, one epilogue quad (hipcc
        // otherwise bunches a tile's re-fills and the previous tile's whole epilogue behind the first MFMAs of the tile)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // VALU
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // VALU
        // The clamped conversions are inline asm: no group mask matches them, so they float -- and hipcc sinks them to ONE wait state in front of the
        // MFMA that reads their result (a VALU write the hazard recogniser cannot see needs 2: tests/test_host_cpu.py
        // test_asm_conversions_are_two_wait_states_ahead_of_the_mfma_that_reads_them caught a carried tile's conversions there).  A scheduling barrier per
        // k-step keeps every conversion in the k-step the source puts it in, k-steps away from its first reader.
        if (tune::kShadeKstepFence) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kept(m)) {
      keepA[kKeepAll ? m : 0] = accA;
      keepB[kKeepAll ? m : 0] = accB;
    }
    if ((ABL & 8) && !kept(m)) {
      asm volatile("" ::"v"(accA), "v"(accB));
#pragma unroll
      for (int g = 0; g < 8; ++g) asm volatile("" : "=v"(outA[8 * m + g]), "=v"(outB[8 * m + g]));
    } else if (m + 1 < MT) {
      pA = accA;
      pB = accB;
    } else if (!kept(m)) {
      if (CARRY_OUT && kCarry) {
        pend.a = accA;
        pend.b = accB;
      } else {
        gA = mfma_guard<ET, RELU>(accA);
        gB = mfma_guard<ET, RELU>(accB);
#pragma unroll
        for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(accA, m, g, outA, gA);
#pragma unroll
        for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(accB, m, g, outB, gB);
      }
    }
  }
}

// A5+A6, 16-bit MFMA path, two sample blocks per wave (layer_16x2): workgroup = 4 waves (one per SIMD) x 64 samples = the same
// 256-sample tile as shade_mlp16_kernel; same weight stream, ring and bias blocks.
template <class ET, int FP, int FD>
__global__ __launch_bounds__(256) void shade_mlp16x2_kernel(ShadeArgs a) {
  static_assert(FP == 10 && FD == 4, "fragment positions below assume the 10-4 shading encoding");
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD);
  constexpr int WAVES = 4, TILE = WAVES * 64, CF = tune::kChunkFrags2, RS = tune::kRingSlots2, LPW = CF / WAVES;
  constexpr int kRingBytes = CF * RS * 1024;
  static_assert(CF % WAVES == 0 && CF % tune::kRegFrags2 == 0 && kShadeFrags16 % CF == 0 && CF % 8 == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW, tune::kRegFrags2> WS;
  constexpr int kStashPerBlock = (QP / 8 + QD / 8) * 1024;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kShadeBiasFloats * 4 + WAVES * 2 * kStashPerBlock];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const uint32_t stash = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + kShadeBiasFloats * 4 +
                         wave * 2 * kStashPerBlock + lane * 16;
  int total = a.total ? *a.total : a.max_samples;
  if (total > a.max_samples) total = a.max_samples;
  const int ntiles = (total + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;    // workgroup-uniform
  {
    float* lds_bias = reinterpret_cast<float*>(lds + kRingBytes);
    for (int i = threadIdx.x; i < kShadeBiasFloats; i += blockDim.x) lds_bias[i] = a.net.bias[i];
  }
  __syncthreads();
  WS st;
  ws_start(st, a.net.w, kShadeFrags16 * 1024, lds, wave, lane);
  const uint32_t bias0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + h * 64;
  const uint32_t* bo = a.net.b_off;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s0 = tile * TILE + wave * 64 + j, s1 = s0 + 32;
    uint32_t hA0[64], hB0[64], hA1[64], hB1[64];
    PendingTile2 pend;      // a layer's last tile, converted under the next layer's first MFMAs (tune::kShadeCarry)
    {
      float x[3], dpe[3];
      uint32_t pts0[QP / 2], pts1[QP / 2], dirs[QD / 2];
      load_sample(a, s0, total, x, dpe);
      pe_pack<ET, FP>(x, h, pts0);
      pe_pack<ET, FD>(dpe, h, dirs);
      lds_stash_write<QP / 8>(stash, pts0);
      lds_stash_write<QD / 8>(stash + (QP / 8) * 1024, dirs);
      load_sample(a, s1, total, x, dpe);
      pe_pack<ET, FP>(x, h, pts1);
      pe_pack<ET, FD>(dpe, h, dirs);
      lds_stash_write<QP / 8>(stash + kStashPerBlock, pts1);
      lds_stash_write<QD / 8>(stash + kStashPerBlock + (QP / 8) * 1024, dirs);
      layer_16x2<ET, WS, QP / 8, 0, 8, true, 0, -1, -1, true, true>(st, bias0 + bo[0] * 4, pts0, pts0, pts1, pts1, hA0, hA1, pend);
    }
#pragma unroll 1
    for (int l = 1; l <= 3; l += 2) {
      layer_16x2<ET, WS, 16, 0, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[l] * 4, hA0, hA0, hA1, hA1, hB0, hB1, pend, hA0, hA1);
      layer_16x2<ET, WS, 16, 0, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[l + 1] * 4, hB0, hB0, hB1, hB1, hA0, hA1, pend, hB0, hB1);
    }
    {
      uint32_t pts0[QP / 2], pts1[QP / 2];
      lds_stash_read<QP / 8>(stash, pts0);
      lds_stash_read<QP / 8>(stash + kStashPerBlock, pts1);
      layer_16x2<ET, WS, QP / 8, 16, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[5] * 4, pts0, hA0, pts1, hA1, hB0, hB1, pend, hA0, hA1);   // cat([pts, h])
    }
    layer_16x2<ET, WS, 16, 0, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[6] * 4, hB0, hB0, hB1, hB1, hA0, hA1, pend, hB0, hB1);
    layer_16x2<ET, WS, 16, 0, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[7] * 4, hA0, hA0, hA1, hA1, hB0, hB1, pend, hA0, hA1);
    f32x16 alpha0, alpha1;
    // feature (+alpha row): tile 8 is kept, tile 7 is converted under it -- nothing to carry
    layer_16x2<ET, WS, 16, 0, 9, false, 0, 8, 7, true, false>(st, bias0 + bo[8] * 4, hB0, hB0, hB1, hB1, hA0, hA1, pend, hB0, hB1, &alpha0, &alpha1);
    {
      uint32_t d0[QD / 2], d1[QD / 2];
      lds_stash_read<QD / 8>(stash + (QP / 8) * 1024, d0);
      lds_stash_read<QD / 8>(stash + kStashPerBlock + (QP / 8) * 1024, d1);
      layer_16x2<ET, WS, 16, QD / 8, 4, true, (32 + 4 * 128 + 160 + 2 * 128 + 144) % CF>(st, bias0 + bo[9] * 4, hA0, d0, hA1, d1, hB0, hB1, pend);   // cat([feature, dir])
    }
    f32x16 rgb0, rgb1;
    // (the view layer's last tile is NOT carried into the 8-k-step colour layer: hipcc sinks the conversions to one wait state in front of the MFMAs
    // that read them -- tests/test_host_cpu.py test_asm_conversions_are_two_wait_states_ahead_of_the_mfma_that_reads_them)
    layer_16x2<ET, WS, 8, 0, 1, false, (32 + 4 * 128 + 160 + 2 * 128 + 144 + 72) % CF, 0>(st, bias0 + bo[10] * 4, hB0, hB0, hB1, hB1, hA0, hA1, pend, nullptr, nullptr, &rgb0, &rgb1);
    if (h == 0 && s0 < total)
      store_raw(a, s0, rgb0[0], rgb0[1], rgb0[2], alpha0[0]);
    if (h == 0 && s1 < total)
      store_raw(a, s1, rgb1[0], rgb1[1], rgb1[2], alpha1[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// fp32 parity mode of the shading net: same structure on the fp32 MFMA engine, accurate sincos.
template <int FP, int FD>
__global__ __launch_bounds__(256) void shade_mlp32_kernel(ShadeArgs a) {
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD);
  constexpr int TILE = 4 * 32;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  int total = a.total ? *a.total : a.max_samples;
  if (total > a.max_samples) total = a.max_samples;
  const u32x4* w = a.net.w;
  const float* b = a.net.bias;

  for (int tile = blockIdx.x; tile * TILE < total; tile += gridDim.x) {
    const int s = tile * TILE + wave * 32 + j;
    if (tile * TILE + wave * 32 >= total) continue;
    // the weight addresses do not depend on the tile: without this the compiler hoists every A-fragment
    // load out of the tile loop (loop-invariant code motion) and spills thousands of registers
    asm volatile("" : "+v"(w), "+v"(b));
    float x[3], dpe[3];
    load_sample(a, s, total, x, dpe);
    float pts[QP], dirs[QD], hA[144], hB[128];      // hA also receives the 9-tile feature(+alpha) layer
    pe_eval<FP, true>(x, h, pts);
    pe_eval<FD, true>(dpe, h, dirs);
    layer_f32<QP, 0, 8, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, pts, pts, hA);
#pragma unroll 1
    for (int l = 1; l <= 3; l += 2) {
      layer_f32<128, 0, 8, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, hA, hA, hB);
      layer_f32<128, 0, 8, true>(w + a.net.w_off[l + 1], b + a.net.b_off[l + 1], lane, hB, hB, hA);
    }
    layer_f32<QP, 128, 8, true>(w + a.net.w_off[5], b + a.net.b_off[5], lane, pts, hA, hB);       // cat([pts, h])
    layer_f32<128, 0, 8, true>(w + a.net.w_off[6], b + a.net.b_off[6], lane, hB, hB, hA);
    layer_f32<128, 0, 8, true>(w + a.net.w_off[7], b + a.net.b_off[7], lane, hA, hA, hB);
    layer_f32<128, 0, 9, false>(w + a.net.w_off[8], b + a.net.b_off[8], lane, hB, hB, hA);         // feature (+alpha row)
    const float alpha = hA[128];
    layer_f32<128, QD, 4, true>(w + a.net.w_off[9], b + a.net.b_off[9], lane, hA, dirs, hB);       // cat([feature, dir])
    float rgb[16];
    layer_f32<64, 0, 1, false>(w + a.net.w_off[10], b + a.net.b_off[10], lane, hB, hB, rgb);
    if (h == 0 && s < total)
      *reinterpret_cast<float4*>(a.raw_out + static_cast<size_t>(s) * 4) = make_float4(rgb[0], rgb[1], rgb[2], alpha);
  }
}

// Debug/parity: explicit shading-net input features in the reference's column order.
static __global__ __launch_bounds__(256) void shade_features_kernel(ShadeArgs a, float* feat, int FP, int FD) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.max_samples) return;
  float x[3], dpe[3];
  load_sample(a, s, a.max_samples, x, dpe);
  const int NP = 3 + 6 * FP, ND = 3 + 6 * FD;
  float* f = feat + static_cast<size_t>(s) * (NP + ND);
  for (int c = 0; c < 3; ++c) {
    f[c] = x[c];
    f[NP + c] = dpe[c];
  }
  for (int b = 0; b < FP; ++b)
    for (int c = 0; c < 3; ++c) {
      float sn, co;
      sincosf(x[c] * static_cast<float>(1 << b), &sn, &co);
      f[3 + 6 * b + c] = sn;
      f[3 + 6 * b + 3 + c] = co;
    }
  for (int b = 0; b < FD; ++b)
    for (int c = 0; c < 3; ++c) {
      float sn, co;
      sincosf(dpe[c] * static_cast<float>(1 << b), &sn, &co);
      f[NP + 3 + 6 * b + c] = sn;
      f[NP + 3 + 6 * b + 3 + c] = co;
    }
}

}  // namespace adanerf
