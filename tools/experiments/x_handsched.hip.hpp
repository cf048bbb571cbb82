// EXPERIMENT (compiled only with -DADN_EXPERIMENT, tools/ablate.sh): hand-scheduled forms of layer_16 (HsLayer) and
// layer_16x3 (HsLayer3) -- every LDS read, every s_waitcnt and the place of every epilogue instruction chosen by hand.
// Measured in round 2 (profiles/r02_handsched.md): bit-identical results, deterministic, but no faster than what hipcc
// schedules from the plain C++ layers (shading kernel +0.5 ... 1.4 %, split-precision sampling kernel 13-17 % slower), so the
// shipped kernels use the compiler-scheduled layers.  Kept because the timing ablations are only additive in this form and
// because of what it documents: the hazards a hand-scheduled MFMA stream has to cover itself (see tuning.hpp kHsStepNop,
// tools/check_inflight.py, tools/probes/determinism.py).
// Included twice: from k_mlp16.hip.hpp (part 1, HsLayer) and from k_sampling16.hip.hpp (part 2, HsLayer3).
#ifndef ADN_HANDSCHED_PART2
#ifndef ADANERF_X_HANDSCHED_1
#define ADANERF_X_HANDSCHED_1
namespace adanerf {

// ------------------------------------------------------------------------------------------
// Hand-scheduled form of layer_16 (tune::kHandSched).  hipcc schedules the plain-C++ fragment re-fills of layer_16 as it
// likes: in the ISA of the shipped round-2 kernel the "register ring" had collapsed into bursts of ds_read_b128 followed by
// s_waitcnt lgkmcnt(0) right behind a read (the whole LDS latency exposed in the MFMA chain), the bias block of the next
// tile was read and waited for with lgkmcnt(0) between the last two MFMAs of a tile, and every tile's epilogue sat behind
// s_nop 11 directly after its last MFMA.  Here every LDS access of the layer is issued by hand and every wait is counted:
//   step (m, s):  s_waitcnt lgkmcnt(X) ; v_mfma acc[m & 1] ; ds_read_b128 R[f % NR] <- fragment f + NR       (one asm block)
//   after step (m, 2):  epilogue of tile m - 1 (its accumulator is threaded through that asm block as a "+v" operand, so the
//                       VALU reads cannot be scheduled above it: >= 2 dependent MFMAs behind the MFMA that wrote it, no
//                       s_nop), then the 4 bias reads of tile m + 1 into the accumulator the epilogue just released
//   X = number of LDS reads issued BEHIND the one that is needed (LDS returns in order, so "at most X outstanding" means the
//       needed one has landed): NR - 1 younger fragments (+ 4 where the bias reads of the next tile were issued in between);
//       at s = 0 additionally the tile's own bias reads must have landed.  Counting too few is the only unsafe direction,
//       and LDS / scalar loads the compiler issues itself can only add younger operations.
// The fragment registers and the bias-initialised accumulator are "in flight" between two asm blocks; the compiler sees
// them as ordinary values there, so ws_settle() drains them in front of every loop back-edge (a copy inserted for a phi
// would otherwise read a register the LDS has not written yet); tools/asm_report.py --inflight checks the ISA for that.
constexpr int hs_epilogue_step(int KS) { return KS >= 4 ? 2 : KS - 1; }
constexpr int hs_wait(int M, int S, int KS, int MT, int NR) {
  const int g = M * KS + S, SB = hs_epilogue_step(KS);
  int cnt = (g <= NR - 1) ? 1 : 0;                      // bias of tile 0, issued in front of step 0
  for (int m = 0; m + 1 < MT; ++m) {
    const int gb = m * KS + SB;                         // bias of tile m + 1, issued behind step (m, SB)
    if (g - NR <= gb && gb <= g - 1) ++cnt;
  }
  return NR - 1 + 4 * cnt;
}
// step (M, 0): LDS reads issued behind the tile's own bias reads (= re-fills of the steps in between)
// (if that is more than the step's own count, the bias is older than the fragment the step waits for: use that count)
constexpr int hs_wait_bias(int M, int KS, int MT, int NR) {
  const int younger = (M == 0) ? 0 : (M * KS - 1) - ((M - 1) * KS + hs_epilogue_step(KS));
  const int x = hs_wait(M, 0, KS, MT, NR);
  return younger < x ? younger : x;
}

template <class ET>
struct MfmaText;
template <>
struct MfmaText<Bf16> {
  static constexpr bool kBf16 = true;
};
template <>
struct MfmaText<Fp16> {
  static constexpr bool kBf16 = false;
};

// wait, MFMA, re-fill -- one block, so nothing is scheduled in between
template <class ET, int WAIT, int OFF>
__device__ __forceinline__ void hs_step_asm(f32x16& acc, u32x4& r, const u32x4& b, uint32_t addr) {
  if (tune::kAblateShade & 2) {       // timing ablation: no LDS re-fill (wrong results)
    if (MfmaText<ET>::kBf16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc), "+v"(r) : "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc), "+v"(r) : "v"(b));
    return;
  }
  if (MfmaText<ET>::kBf16)
    asm volatile("s_nop %6\n\ts_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%5"
                 : "+v"(acc), "+v"(r)
                 : "v"(b), "v"(addr), "n"(tune::kHsWaitZero ? 0 : WAIT), "n"(OFF), "n"(tune::kHsStepNop));
  else
    asm volatile("s_nop %6\n\ts_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%5"
                 : "+v"(acc), "+v"(r)
                 : "v"(b), "v"(addr), "n"(tune::kHsWaitZero ? 0 : WAIT), "n"(OFF), "n"(tune::kHsStepNop));
}
// the same with a second accumulator named as an operand: whatever reads it is ordered behind this block
// NOPS: extra wait states for tiles too short to put two dependent MFMAs between the producer and this block
template <class ET, int WAIT, int OFF, int NOPS>
__device__ __forceinline__ void hs_step_asm_thread(f32x16& acc, u32x4& r, const u32x4& b, uint32_t addr, f32x16& prev) {
  if (MfmaText<ET>::kBf16)
    asm volatile("s_nop %8\n\ts_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %0\n\tds_read_b128 %1, %4 offset:%6\n\ts_nop %7"
                 : "+v"(acc), "+v"(r), "+v"(prev)
                 : "v"(b), "v"(addr), "n"(tune::kHsWaitZero ? 0 : WAIT), "n"(OFF), "n"(NOPS), "n"(tune::kHsStepNop));
  else
    asm volatile("s_nop %8\n\ts_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %3, %0\n\tds_read_b128 %1, %4 offset:%6\n\ts_nop %7"
                 : "+v"(acc), "+v"(r), "+v"(prev)
                 : "v"(b), "v"(addr), "n"(tune::kHsWaitZero ? 0 : WAIT), "n"(OFF), "n"(NOPS), "n"(tune::kHsStepNop));
}
// the tile's bias block has landed (X younger reads may still be in flight): it becomes the accumulator
template <int X>
__device__ __forceinline__ void hs_bias_take(BiasRegs& r, f32x16* acc) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r.b0), "+v"(r.b1), "+v"(r.b2), "+v"(r.b3) : "n"(tune::kHsWaitZero ? 0 : X));
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

template <class WS>
__device__ __forceinline__ void ws_settle(WS& st) {
  static_assert(WS::kRegs == 4 || WS::kRegs == 8, "register ring depth");
  if (WS::kRegs == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st.R[0]), "+v"(st.R[1]), "+v"(st.R[2]), "+v"(st.R[3]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(st.R[0]), "+v"(st.R[1]), "+v"(st.R[2]), "+v"(st.R[3]), "+v"(st.R[4 % WS::kRegs]), "+v"(st.R[5 % WS::kRegs]),
                   "+v"(st.R[6 % WS::kRegs]), "+v"(st.R[7 % WS::kRegs]));
}

template <class ET, class WS, int S1, int S2, int MT, bool RELU, int FPOS, int KEEP_F32_TILE>
struct HsLayer {
  static constexpr int CF = WS::kChunk, NR = WS::kRegs, KS = S1 + S2;
  static constexpr int SB = hs_epilogue_step(KS);
  static_assert(KS >= 2 && CF % NR == 0, "tile length / ring depth");

  // the part of a tile's epilogue that reads the accumulator
  template <int M>
  static __device__ __forceinline__ void finish(f32x16& acc, uint32_t* out, f32x16* keep) {
    // the empty asm statements pin the results here (asm volatile statements keep their order): without them the compiler
    // sinks the conversion towards its first use and the accumulator stays live over the following tiles
    if ((tune::kAblateShade & 8) && KEEP_F32_TILE == -1) {      // timing ablation: no epilogue (wrong results)
      uint32_t* o = out + 8 * M;
      asm volatile("" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]) : "v"(acc));
    } else if (KEEP_F32_TILE == kKeepAllF32) {
      keep[M] = acc;
      asm volatile("" : "+v"(keep[M]));
    } else if (KEEP_F32_TILE == M) {
      *keep = acc;
      asm volatile("" : "+v"(*keep));
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(acc, M, g, out);
      uint32_t* o = out + 8 * M;
      asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));
    }
  }

  template <int M, int S>
  static __device__ __forceinline__ void step(WS& st, uint32_t bias_addr, const uint32_t* in1, const uint32_t* in2, uint32_t* out,
                                              f32x16* keep, f32x16 (&acc)[2], BiasRegs& br) {
    constexpr int g = M * KS + S;
    constexpr int f = (FPOS + g) % CF;
    ws_position<tune::kAblateShade>(st, f, false);           // no MFMA -> VALU padding needed: the epilogues are ordered by their operands
    constexpr int q = f + NR;
    const uint32_t addr = (q < CF) ? st.rd_cur : st.rd_next;
    constexpr int off = ((q < CF) ? q : q - CF) * 1024;
    constexpr int wait = hs_wait(M, S, KS, MT, NR);
    const uint32_t* src = (S < S1) ? (in1 + 4 * S) : (in2 + 4 * (S - S1));
    const u32x4 b = {src[0], src[1], src[2], src[3]};
    if (S == 0) hs_bias_take<hs_wait_bias(M, KS, MT, NR)>(br, &acc[M & 1]);
    if (S == SB && M >= 1) hs_step_asm_thread<ET, wait, off, (KS >= 4 ? 0 : 11)>(acc[M & 1], st.R[f % NR], b, addr, acc[(M + 1) & 1]);
    else hs_step_asm<ET, wait, off>(acc[M & 1], st.R[f % NR], b, addr);
    if (S == SB) {
      if (M >= 1) finish<(M >= 1 ? M - 1 : 0)>(acc[(M + 1) & 1], out, keep);
      if (M + 1 < MT && !(tune::kAblateShade & 4)) lds_bias_issue(bias_addr + (M + 1) * 128, br);
    }
  }
  template <int M, int... S>
  static __device__ __forceinline__ void tile(WS& st, uint32_t bias_addr, const uint32_t* in1, const uint32_t* in2, uint32_t* out,
                                              f32x16* keep, f32x16 (&acc)[2], BiasRegs& br, std::integer_sequence<int, S...>) {
    (step<M, S>(st, bias_addr, in1, in2, out, keep, acc, br), ...);
  }
  template <int... M>
  static __device__ __forceinline__ void tiles(WS& st, uint32_t bias_addr, const uint32_t* in1, const uint32_t* in2, uint32_t* out,
                                               f32x16* keep, f32x16 (&acc)[2], BiasRegs& br, std::integer_sequence<int, M...>) {
    (tile<M>(st, bias_addr, in1, in2, out, keep, acc, br, std::make_integer_sequence<int, KS>{}), ...);
  }
  static __device__ __forceinline__ void run(WS& st, uint32_t bias_addr, const uint32_t* in1, const uint32_t* in2, uint32_t* out,
                                             f32x16* keep) {
    f32x16 acc[2];
    asm volatile("s_nop 1");                 // VALU-written B operands of the first steps (the compiler cannot see the MFMAs)
    BiasRegs br;
    if (tune::kAblateShade & 4) br.b0 = br.b1 = br.b2 = br.b3 = f32x4{0.f, 0.f, 0.f, 0.f};      // timing ablation: no bias reads
    else lds_bias_issue(bias_addr, br);
    tiles(st, bias_addr, in1, in2, out, keep, acc, br, std::make_integer_sequence<int, MT>{});
    // last tile: nothing follows that could carry its accumulator; 14 wait states behind its last MFMA (11 required)
    asm volatile("s_nop 13" : "+v"(acc[(MT - 1) & 1]));
    finish<MT - 1>(acc[(MT - 1) & 1], out, keep);
  }
};

}  // namespace adanerf
#endif
#else
#ifndef ADANERF_X_HANDSCHED_2
#define ADANERF_X_HANDSCHED_2
namespace adanerf {

// ------------------------------------------------------------------------------------------
// Hand-scheduled form of layer_16x3 (tune::kHandSched; see HsLayer in k_mlp16.hip.hpp for the rules).  One wave per SIMD:
// whatever this wave's instruction stream exposes, the MFMA pipe idles for.  Per k-step one asm block
//   s_waitcnt lgkmcnt(X) ; acc += Whi.xhi ; cross += Whi.xlo' ; cross += Wlo'.xhi ; re-fill the (hi, lo') fragment pair
// with X counted (never 0 inside a tile), the first block of a tile starting the cross chain from the constant 0, the
// previous tile's epilogue spread one accumulator pair per k-step BEHIND the steps 1 .. (the accumulators are operands of
// those blocks, so a pair can be scheduled neither above its block nor -- results pinned -- below the next one), and the
// bias block of the next tile read into the accumulator the epilogue has just released.
constexpr int hs3_pairs_per_step(int KS) { return (8 + (KS - 1) - 1) / (KS - 1); }
constexpr int hs3_last_epilogue_step(int KS) { return (8 + hs3_pairs_per_step(KS) - 1) / hs3_pairs_per_step(KS); }   // pairs sit behind steps 1 .. E
constexpr int hs3_wait(int M, int S, int KS, int MT, int NRP) {
  const int g = M * KS + S, E = hs3_last_epilogue_step(KS);
  int cnt = (g <= NRP - 1) ? 1 : 0;                     // bias of tile 0, issued in front of step 0
  for (int m = 0; m + 1 < MT; ++m) {
    const int gb = m * KS + E;                          // bias of tile m + 1, issued behind step (m, E)
    if (g - NRP <= gb && gb <= g - 1) ++cnt;
  }
  return 2 * (NRP - 1) + 4 * cnt;
}
constexpr int hs3_wait_bias(int M, int KS, int MT, int NRP) {
  const int younger = (M == 0) ? 0 : 2 * ((M * KS - 1) - ((M - 1) * KS + hs3_last_epilogue_step(KS)));
  const int x = hs3_wait(M, 0, KS, MT, NRP);
  return younger < x ? younger : x;
}

// The MFMA pipe of a SIMD executes one MFMA at a time and a wave's next MFMA does not issue before the pipe is free, so
// whatever follows an MFMA in program order runs in THAT MFMA's 32-cycle shadow only: three MFMAs back to back with the
// step's VALU work behind them (first version of this layer: 1.62 ms against 1.39 for the compiler's schedule) leave two
// shadows empty.  A k-step is therefore three blocks, each followed by a third of the previous tile's epilogue pair:
//   A: s_nop ; s_waitcnt lgkmcnt(X) ; acc += Whi.xhi          | v = acc' + cross' / 2048, ReLU, hi = f16(v)
//   B: cross += Whi.xlo'                                      | r = v - f32(hi)
//   C: cross += Wlo'.xhi ; re-fill the (hi, lo') pair         | lo' = f16(2048 r)
// (acc', cross' = previous tile; they are operands of block A, the intermediate values of blocks B and C, so no part can be
// scheduled above its block, and the results are pinned below.)
// Register files: one wave per SIMD owns 512 registers, 256 architectural VGPRs + 256 AccVGPRs.  MFMA sources and LDS
// destinations may be either, VALU results and inline-asm "v" operands only the former.  The fragment ring and the lo'
// activation sets are therefore pinned to AccVGPRs ("a" operands): left to the allocator they were copied back and forth
// (v_accvgpr_read / _write) in front of the blocks -- including fragment registers an LDS read was still in flight to.
template <int WAIT, bool THREAD>
__device__ __forceinline__ void hs3_block_a(f32x16& acc, const u32x4& rh, const u32x4& bh, f32x16& pacc, f32x16& pcross) {
  if (THREAD)
    asm volatile("s_nop %6\n\ts_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_32x32x16_f16 %0, %3, %4, %0"
                 : "+v"(acc), "+v"(pacc), "+v"(pcross)
                 : "a"(rh), "v"(bh), "n"(tune::kHsWaitZero ? 0 : WAIT), "n"(tune::kHsStepNop));
  else
    asm volatile("s_nop %4\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
                 : "+v"(acc)
                 : "a"(rh), "v"(bh), "n"(tune::kHsWaitZero ? 0 : WAIT), "n"(tune::kHsStepNop));
}
template <bool FIRST, bool THREAD>
__device__ __forceinline__ void hs3_block_b(f32x16& cross, const u32x4& rh, const u32x4& bl, float& t0, float& t1) {
  // s_nop: the lo' operands are AccVGPR tuples the compiler assembles with v_accvgpr_write right in front of the block
  if (FIRST) asm volatile("s_nop %3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(cross) : "a"(rh), "a"(bl), "n"(tune::kHsStepNop));
  else if (THREAD)
    asm volatile("s_nop %5\n\tv_mfma_f32_32x32x16_f16 %0, %3, %4, %0" : "+v"(cross), "+v"(t0), "+v"(t1) : "a"(rh), "a"(bl), "n"(tune::kHsStepNop));
  else asm volatile("s_nop %3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cross) : "a"(rh), "a"(bl), "n"(tune::kHsStepNop));
}
template <int OFF, bool THREAD>
__device__ __forceinline__ void hs3_block_c(f32x16& cross, u32x4& rh, u32x4& rl, const u32x4& bh, uint32_t addr, float& t0, float& t1) {
  if (THREAD)
    asm volatile("s_nop %9\n\tv_mfma_f32_32x32x16_f16 %0, %2, %5, %0\n\tds_read_b128 %1, %6 offset:%7\n\tds_read_b128 %2, %6 offset:%8"
                 : "+v"(cross), "+a"(rh), "+a"(rl), "+v"(t0), "+v"(t1)
                 : "v"(bh), "v"(addr), "n"(OFF), "n"(OFF + 1024), "n"(tune::kHsStepNop));
  else
    asm volatile("s_nop %7\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6"
                 : "+v"(cross), "+a"(rh), "+a"(rl)
                 : "v"(bh), "v"(addr), "n"(OFF), "n"(OFF + 1024), "n"(tune::kHsStepNop));
}
// no LDS read in flight to the (AccVGPR) fragment ring
template <class WS>
__device__ __forceinline__ void ws_settle_acc(WS& st) {
  static_assert(WS::kRegs == 8, "register ring depth");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+a"(st.R[0]), "+a"(st.R[1]), "+a"(st.R[2]), "+a"(st.R[3]), "+a"(st.R[4]), "+a"(st.R[5]), "+a"(st.R[6]), "+a"(st.R[7]));
}

template <class WS, int KS, int MT, bool LAST, int FPOS>
struct HsLayer3 {
  static constexpr int CF = WS::kChunk, NR = WS::kRegs, NRP = NR / 2;
  static constexpr int PER = hs3_pairs_per_step(KS), E = hs3_last_epilogue_step(KS);
  static_assert(KS >= 2 && NR % 2 == 0 && CF % NR == 0 && E <= KS - 1 && 2 * NRP + 4 <= 15, "tile length / ring depth");

  // the three parts of one accumulator pair's epilogue (values 2 pi, 2 pi + 1 of tile M); v0 / v1 carry the state
  template <int M>
  static __device__ __forceinline__ void part_a(const f32x16& acc, const f32x16& cross, int pi, uint32_t* out_hi, float* out_f32, float& v0,
                                                float& v1) {
    v0 = __builtin_fmaf(cross[2 * pi], 1.0f / kSplitScale, acc[2 * pi]);
    v1 = __builtin_fmaf(cross[2 * pi + 1], 1.0f / kSplitScale, acc[2 * pi + 1]);
    if (LAST) {
      out_f32[16 * M + 2 * pi] = v0;
      out_f32[16 * M + 2 * pi + 1] = v1;
      asm volatile("" : "+v"(out_f32[16 * M + 2 * pi]), "+v"(out_f32[16 * M + 2 * pi + 1]));
    } else {
      v0 = relu_bits(v0);
      v1 = relu_bits(v1);
      f32x2 v = {v0, v1};
      out_hi[8 * M + pi] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
      asm volatile("" : "+v"(out_hi[8 * M + pi]), "+v"(v0), "+v"(v1));
    }
  }
  template <int M>
  static __device__ __forceinline__ void part_b(int pi, const uint32_t* out_hi, float& v0, float& v1) {
    if (LAST) return;
    const f32x2 hf = __builtin_convertvector(__builtin_bit_cast(f16x2, out_hi[8 * M + pi]), f32x2);
    v0 = v0 - hf[0];
    v1 = v1 - hf[1];
    asm volatile("" : "+v"(v0), "+v"(v1));
  }
  template <int M>
  static __device__ __forceinline__ void part_c(int pi, uint32_t* out_lo, float& v0, float& v1) {
    if (LAST) return;
    f32x2 r = {v0 * kSplitScale, v1 * kSplitScale};
    out_lo[8 * M + pi] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
    asm volatile("" : "+v"(out_lo[8 * M + pi]));
  }

  template <int M, int S>
  static __device__ __forceinline__ void step(WS& st, uint32_t bias_addr, const uint32_t* in_hi, const uint32_t* in_lo, uint32_t* out_hi,
                                              uint32_t* out_lo, float* out_f32, f32x16 (&acc)[2], f32x16 (&cross)[2], BiasRegs& br) {
    constexpr int g = M * KS + S;
    constexpr int f = (FPOS + 2 * g) % CF;               // even
    ws_position<tune::kAblateSample>(st, f, false);
    constexpr int q = f + NR;
    const uint32_t addr = (q < CF) ? st.rd_cur : st.rd_next;
    constexpr int off = ((q < CF) ? q : q - CF) * 1024;
    constexpr int wait = hs3_wait(M, S, KS, MT, NRP);
    const u32x4 bh = {in_hi[4 * S], in_hi[4 * S + 1], in_hi[4 * S + 2], in_hi[4 * S + 3]};
    const u32x4 bl = {in_lo[4 * S], in_lo[4 * S + 1], in_lo[4 * S + 2], in_lo[4 * S + 3]};
    constexpr int cur = M & 1, prv = (M + 1) & 1;
    constexpr bool EPI = M >= 1 && S >= 1 && S <= E && !(tune::kAblateSample & 8);      // this step carries epilogue pairs of tile M - 1
    constexpr int PM = M >= 1 ? M - 1 : 0;
    u32x4& rh = st.R[f % NR];
    u32x4& rl = st.R[(f + 1) % NR];
    float v0[PER], v1[PER];
    if (S == 0) hs_bias_take<hs3_wait_bias(M, KS, MT, NRP)>(br, &acc[cur]);
    hs3_block_a<wait, EPI>(acc[cur], rh, bh, acc[prv], cross[prv]);
    if (EPI) {
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if ((S - 1) * PER + k < 8) part_a<PM>(acc[prv], cross[prv], (S - 1) * PER + k, out_hi, out_f32, v0[k], v1[k]);
    }
    hs3_block_b<S == 0, EPI && !LAST>(cross[cur], rh, bl, v0[0], v1[0]);
    if (EPI) {
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if ((S - 1) * PER + k < 8) part_b<PM>((S - 1) * PER + k, out_hi, v0[k], v1[k]);
    }
    hs3_block_c<off, EPI && !LAST>(cross[cur], rh, rl, bh, addr, v0[0], v1[0]);
    if (EPI) {
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if ((S - 1) * PER + k < 8) part_c<PM>((S - 1) * PER + k, out_lo, v0[k], v1[k]);
    }
    if (S == E && M + 1 < MT) lds_bias_issue(bias_addr + (M + 1) * 128, br);
  }
  template <int M, int... S>
  static __device__ __forceinline__ void tile(WS& st, uint32_t bias_addr, const uint32_t* in_hi, const uint32_t* in_lo, uint32_t* out_hi,
                                              uint32_t* out_lo, float* out_f32, f32x16 (&acc)[2], f32x16 (&cross)[2], BiasRegs& br,
                                              std::integer_sequence<int, S...>) {
    (step<M, S>(st, bias_addr, in_hi, in_lo, out_hi, out_lo, out_f32, acc, cross, br), ...);
  }
  template <int... M>
  static __device__ __forceinline__ void tiles(WS& st, uint32_t bias_addr, const uint32_t* in_hi, const uint32_t* in_lo, uint32_t* out_hi,
                                               uint32_t* out_lo, float* out_f32, f32x16 (&acc)[2], f32x16 (&cross)[2], BiasRegs& br,
                                               std::integer_sequence<int, M...>) {
    (tile<M>(st, bias_addr, in_hi, in_lo, out_hi, out_lo, out_f32, acc, cross, br, std::make_integer_sequence<int, KS>{}), ...);
  }
  static __device__ __forceinline__ void run(WS& st, uint32_t bias_addr, const uint32_t* in_hi, const uint32_t* in_lo, uint32_t* out_hi,
                                             uint32_t* out_lo, float* out_f32) {
    f32x16 acc[2], cross[2];
    BiasRegs br;
    lds_bias_issue(bias_addr, br);
    tiles(st, bias_addr, in_hi, in_lo, out_hi, out_lo, out_f32, acc, cross, br, std::make_integer_sequence<int, MT>{});
    constexpr int last = (MT - 1) & 1;
    asm volatile("s_nop 13" : "+v"(acc[last]), "+v"(cross[last]));
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) epilogue_pair_16x3<LAST>(acc[last], cross[last], MT - 1, pi, out_hi, out_lo, out_f32);
  }
};

}  // namespace adanerf
#endif
#endif
