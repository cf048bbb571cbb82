// SURVEY 8f N4, second half: both networks in any exportable topology (sampling net of 2..8 layers, NeRF trunk of 1..8 layers, width
// 64 / 128 / 256, skips at any layers or none, src/models.py:18-82, 199-277) and any encoding layout on the 16-bit MFMA pipe.
// The specialised 8 x 256 kernels stage one fragment stream through an LDS ring whose chunk positions are compile-time
// constants of that topology.  Here the layer table is a run-time argument: a run-time loop over the hidden layers with the layer's
// tiles and k-steps unrolled, and the weights staged per output tile through two LDS buffers shared by the workgroup (TileStage below;
// shipped).  The first form -- every wave fetching its fragments straight from global memory (L2 hits), layer_16_direct /
// layer_16x3_direct -- is kept as the experiment baseline (tuning.hpp kGenericStaged, profiles/r03_generic_staged.md).
// Device code only (gfx950, wave64); part of kernels.hip.hpp (main translation unit: accumulators in architectural VGPRs).
#pragma once
#include "k_generic_f32.hip.hpp"      // GenericTopo
#include "k_mlp16.hip.hpp"
#include "k_sampling16.hip.hpp"   // split_pack, epilogue_pair_16x3

namespace adanerf {

// One 16-bit layer, fragments [m][s][lane][8 x 16 bit] from global memory, bias blocks [m][h][16] fp32 from global memory.
// Input = two register segments of S1 / S2 k-steps (4 packed dwords each); KEEP_F32_TILE as layer_16.
template <class ET, int S1, int S2, int MT, bool RELU, int KEEP_F32_TILE = -1>
__device__ __forceinline__ void layer_16_direct(const u32x4* __restrict__ w, const float* __restrict__ bias, int lane, const uint32_t* in1,
                                                const uint32_t* in2, uint32_t* out, f32x16* keep = nullptr) {
  constexpr int KS = S1 + S2;
  const int h = lane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc;
    const float4* bp = reinterpret_cast<const float4*>(bias + (m * 2 + h) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = bp[g];
      acc[4 * g + 0] = b.x;
      acc[4 * g + 1] = b.y;
      acc[4 * g + 2] = b.z;
      acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 a = w[(m * KS + s) * 64 + lane];
      const uint32_t* src = (s < S1) ? (in1 + 4 * s) : (in2 + 4 * (s - S1));
      const u32x4 b = {src[0], src[1], src[2], src[3]};
      acc = ET::mfma(a, b, acc);
    }
    if (KEEP_F32_TILE == m) {
      *keep = acc;
    } else {
      const int gd = mfma_guard<ET, RELU>(acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(acc, m, g, out, gd);
    }
  }
}

// A5 + A6 for any shading-net topology on the 16-bit engine.  Workgroup = 4 waves x 32 samples (two workgroups per CU).
template <class ET, int FP, int FD, int W>
__global__ __launch_bounds__(256, 2) void shade_mlp16_gen_kernel(ShadeArgs a, GenericTopo t) {
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD), MT = W / 32, KW = W / 16;      // KW: k-steps of a W-wide input
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
    if (tile * TILE + wave * 32 >= total) continue;      // wave-uniform
    asm volatile("" : "+v"(w), "+v"(b));                  // keep the fragment loads inside the loops (see shade_mlp32_kernel)
    float x[3], dpe[3];
    load_sample(a, s, total, x, dpe);
    uint32_t pts[QP / 2], dirs[QD / 2], hA[W / 4], hB[W / 4 + 8];      // hB also receives the (MT + 1)-tile feature (+ alpha) layer's packed part
    pe_pack<ET, FP>(x, h, pts);
    pe_pack<ET, FD>(dpe, h, dirs);
    layer_16_direct<ET, QP / 8, 0, MT, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, pts, pts, hA);
#pragma unroll 1
    for (int l = 1; l < t.depth; ++l) {
      asm volatile("" : "+v"(w), "+v"(b));
      if ((t.cat_mask >> l) & 1) layer_16_direct<ET, QP / 8, KW, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, pts, hA, hB);      // cat([pts, h])
      else layer_16_direct<ET, KW, 0, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, hA, hA, hB);
#pragma unroll
      for (int i = 0; i < W / 4; ++i) hA[i] = hB[i];
    }
    const int lf = t.depth;
    f32x16 alpha_tile;
    layer_16_direct<ET, KW, 0, MT + 1, false, MT>(w + a.net.w_off[lf], b + a.net.b_off[lf], lane, hA, hA, hB, &alpha_tile);        // feature (+ alpha row)
    layer_16_direct<ET, KW, QD / 8, MT / 2, true>(w + a.net.w_off[lf + 1], b + a.net.b_off[lf + 1], lane, hB, dirs, hA);          // cat([feature, dir])
    f32x16 rgb_tile;
    layer_16_direct<ET, KW / 2, 0, 1, false, 0>(w + a.net.w_off[lf + 2], b + a.net.b_off[lf + 2], lane, hA, hA, hB, &rgb_tile);
    if (h == 0 && s < total)
      store_raw(a, s, rgb_tile[0], rgb_tile[1], rgb_tile[2], alpha_tile[0]);
  }
}

// ---- per-tile weight staging through LDS -------------------------------------------------------------------------------------------
// Fetching every A fragment per wave costs 1 KiB through the vector-memory path per MFMA: four SIMDs x 1 KiB per 64 MFMA cycles is the
// whole 64 B / clk of a CU's L1 (profiles/r03_generic_staged.md: the direct form sits at 0.16-0.19 of the MFMA peak).  Staged form: the
// four waves of a workgroup share one copy -- the fragments of output tile t + 1 are DMA-copied global -> LDS (buffer_load ... lds, each wave
// every fourth KiB) while tile t is consumed from the other of two buffers; one wait + barrier per tile says "tile t has landed for
// everybody and everybody is done with tile t - 1".  Tile sizes and addresses are run-time values (the layer table is), so unlike the
// ring of the 8 x 256 kernels (WStream) nothing here is a compile-time stream position.
struct TileStage {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc;   // the network's fragment buffer
#endif
  uint32_t lds_base;   // LDS byte address of buffer 0
  uint32_t lane_off;   // lane * 16
  int wave;            // 0..3 (wave-uniform): copies fragments wave, wave + 4, ...
  uint32_t buf;        // buffer of the tile consumed next
};

template <int BUF_BYTES>
__device__ __forceinline__ void ts_issue(const TileStage& st, uint32_t frag_off, int n_frags, uint32_t buf) {
  // frag_off: first fragment of the tile in 16-byte units (NetParams::w_off units); n_frags: KiB to copy
  for (int i = st.wave; i < n_frags; i += 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t dst = st.lds_base + buf * BUF_BYTES + i * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(dst), 16,
                                             static_cast<int>(st.lane_off), static_cast<int>(frag_off * 16 + i * 1024), 0, 0);
#endif
  }
}

__device__ __forceinline__ void ts_start(TileStage& st, const void* gbase, char* lds, int wave, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gbase), 0, 0x7fffffff, 0x00020000);
#endif
  st.lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
  st.lane_off = lane * 16;
  st.wave = wave;
  st.buf = 0;
}

// the tile issued last has landed and the other buffer is free: returns this lane's read address, starts the copy of the next tile
template <int BUF_BYTES>
__device__ __forceinline__ uint32_t ts_next(TileStage& st, uint32_t next_off, int next_frags) {
  if (tune::kAblateGeneric & 1) asm volatile("s_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  const uint32_t rd = st.lds_base + st.buf * BUF_BYTES + st.lane_off;
  st.buf ^= 1u;
  ts_issue<BUF_BYTES>(st, next_off, next_frags, st.buf);
  return rd;
}

// The instruction order inside a tile is pinned (sched_barrier after every k-step): fragment s + kStageAhead is requested from LDS
// before the MFMAs of k-step s, the bias block of the NEXT tile is requested from global memory while this one computes.  Left to
// itself the scheduler hoisted every LDS read of a tile to its head (92 registers for a 23-step tile) and spilled.
constexpr int kStageAhead = tune::kGenericAhead;

// Keeps the bias loads of a tile loop inside it (hipcc otherwise hoists every one of them out of the loop and spills) WITHOUT taking the
// pointer's address space away: an opaque zero added to the kernel-argument pointer.  Round 3 laundered the pointer itself through an
// asm statement; what came back was a generic pointer, the bias blocks became FLAT loads, and with flat loads in flight hipcc cannot
// count LDS returns: it put `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every tile's first MFMA, which also drained the copy of the
// NEXT tile's weights that had just been issued -- every tile of 4-16 MFMAs paid a global-memory round trip
// (profiles/r04_generic_bias_loads.md).  As global loads they are counted on vmcnt alone and waited for a tile later.
__device__ __forceinline__ int bias_launder() {
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}
__device__ __forceinline__ void bias_request(const float* __restrict__ bias_tile, int h, float (&br)[16]) {
  if (tune::kAblateGeneric & 2) {      // timing ablation: no bias loads at all
#pragma unroll
    for (int r = 0; r < 16; ++r) br[r] = 0.f;
    return;
  }
  const float4* bp = reinterpret_cast<const float4*>(bias_tile + h * 16);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = bp[g];
    br[4 * g + 0] = v.x;
    br[4 * g + 1] = v.y;
    br[4 * g + 2] = v.z;
    br[4 * g + 3] = v.w;
  }
}

// The shading kernel keeps the network's whole bias table in LDS (copied once per workgroup; <= gen_bias_cap<W>() floats: depth + 3
// layers of at most W + 32 outputs) and reads a tile's block [m][h][16] with four ds_read_b128 a tile ahead.  As global loads (until
// round 4) the four 1 KiB-wide requests per tile went through the same vector-memory path as the weight copies -- 4 of the 6 VMEM
// instructions per tile, all 32 lanes of a half fetching the same 64 bytes -- and, worse, made hipcc wait for them with a counted
// vmcnt that also covered the copy of the NEXT tile issued in between (its piece count is a run-time value, so the compiler could not
// leave it in flight): every tile waited for the L2 -> LDS round trip of the tile after it (profiles/r04_generic_bias_loads.md).
template <int W>
constexpr int gen_bias_cap() {
  return 10 * W + 256;
}
// copies the table into LDS with every load of a thread in flight at once (a workgroup of the sampling kernel lives for one 128-ray
// tile: five dependent round trips in front of it would show), then the workgroup barrier
template <int W>
__device__ __forceinline__ void bias_table_fill(float* tab, const float* __restrict__ gbias, uint32_t n_bias) {
  constexpr int PER = (gen_bias_cap<W>() + 255) / 256;
  const int n = min(static_cast<int>(n_bias), gen_bias_cap<W>());      // the host refuses a table beyond the capacity
  float v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = static_cast<int>(threadIdx.x) + 256 * k;
    v[k] = i < n ? gbias[i] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = static_cast<int>(threadIdx.x) + 256 * k;
    if (i < gen_bias_cap<W>()) tab[i] = v[k];
  }
  __syncthreads();
}
__device__ __forceinline__ void bias_request_lds(uint32_t tile_addr, int h, float (&br)[16]) {      // tile_addr: LDS byte address of block [m][0][0]
  typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
  const lds_f32x4_ptr bp = (lds_f32x4_ptr)(uintptr_t)(tile_addr + h * 64);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = bp[g];
    br[4 * g + 0] = v[0];
    br[4 * g + 1] = v[1];
    br[4 * g + 2] = v[2];
    br[4 * g + 3] = v[3];
  }
}

// One 16-bit layer for NB blocks of 32 samples per wave, every fragment read once from LDS and used NB times.
// in1 / in2 / out: NB register arrays of I1 / I2 / O dwords each.  br: bias block of the tile about to run (requested a tile ago).
// (next_off, next_frags, next_bias): first tile of the layer that follows.  bias / next_bias: LDS byte addresses (bias_request_lds).
template <class ET, int NB, int BUF_BYTES, int S1, int S2, int MT, bool RELU, int I1, int I2, int O, int KEEP_F32_TILE = -1, bool LDSB = true>
__device__ __forceinline__ void layer_16_staged(TileStage& st, float (&br)[16], uint32_t w_off, uint32_t bias, int lane,
                                                const uint32_t* in1, const uint32_t* in2, uint32_t* out, uint32_t next_off, int next_frags,
                                                uint32_t next_bias, f32x16* keep = nullptr, const float* __restrict__ gbias = nullptr) {
  constexpr int KS = S1 + S2, D = KS < kStageAhead ? KS : kStageAhead;
  static_assert(KS * 1024 <= BUF_BYTES, "tile does not fit its LDS buffer");
  const int h = lane >> 5;
  // Where do a tile's conversions (ReLU + pack, 16 x NB values) go?  tune::kGenericSpread<W>: spread, a quad at a time, over the k-steps of
  // the NEXT tile, between its MFMAs -- with one wave per SIMD (width 256) nothing else would issue while the pipe works, and nothing
  // else would keep the pipe working while they issue; costs a second accumulator set.  Otherwise right behind the tile's own
  // MFMAs (in one piece behind the next tile's barrier: measured in round 4, no effect, removed).  A layer's last tile always converts at once: the next layer reads its outputs.
  constexpr bool SPREAD = tune::kGenericSpread != 0 && (tune::kGenericSpread >= 2 || O >= 64);      // O = W / 4 (+ 8): 1 = width 256 only
  constexpr int QUADS = 4 * NB, QPS = (QUADS + KS - 1) / KS;      // quads of the previous tile converted per k-step
  f32x16 accs[SPREAD ? 2 : 1][NB];
  int guards[SPREAD ? 2 : 1][NB];      // mfma_guard of each finished accumulator (k_mlp16.hip.hpp): set behind a tile's last k-step
  auto convert_quad = [&](f32x16 (&ac)[NB], int (&gd)[NB], int mm, int q) {
    const int nb = q >> 2, g = q & 3;
    if (KEEP_F32_TILE == mm) {
      if (g == 0) keep[nb] = ac[nb];
    } else {
      epilogue_quad_16<ET, RELU>(ac[nb], mm, g, out + nb * O, gd[nb]);
    }
  };
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bool last = m == MT - 1;
    f32x16 (&acc)[NB] = accs[SPREAD ? (m & 1) : 0];
    f32x16 (&pacc)[NB] = accs[SPREAD ? ((m & 1) ^ 1) : 0];
    int (&gacc)[NB] = guards[SPREAD ? (m & 1) : 0];
    int (&gpacc)[NB] = guards[SPREAD ? ((m & 1) ^ 1) : 0];
    const uint32_t rd = ts_next<BUF_BYTES>(st, last ? next_off : w_off + (m + 1) * KS * 64, last ? next_frags : KS);
    u32x4 fr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) fr[i] = lds_read128(rd + i * 1024);
    if (LDSB && tune::kGenericBiasDirect && !(tune::kAblateGeneric & 2)) {
      // the tile's own bias block straight into its accumulators (no copy a tile ahead, no 16 moves per block): the first MFMA
      // waits for these reads and its first fragment together
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float b16[16];
        bias_request_lds(bias + m * 128, h, b16);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = b16[r];
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = br[r];
      if (tune::kAblateGeneric & 2) {      // timing ablation: no bias reads at all
#pragma unroll
        for (int r = 0; r < 16; ++r) br[r] = 0.f;
      } else if constexpr (LDSB) {
        bias_request_lds(last ? next_bias : bias + (m + 1) * 128, h, br);
      } else {      // table in global memory: bias / next_bias are byte offsets into it
        bias_request(gbias + ((last ? next_bias : bias + (m + 1) * 128) >> 2), h, br);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 a = fr[s % D];
      if (s + D < KS) fr[s % D] = lds_read128(rd + (s + D) * 1024);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint32_t* src = (s < S1) ? (in1 + nb * I1 + 4 * s) : (in2 + nb * I2 + 4 * (s - S1));
        const u32x4 b = {src[0], src[1], src[2], src[3]};
        acc[nb] = ET::mfma(a, b, acc[nb]);
      }
      if (SPREAD && m > 0) {
        if (s == 0 && KEEP_F32_TILE != m - 1) {      // the previous tile's guards, a k-step behind its last MFMAs: no wait states
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) gpacc[nb] = mfma_guard<ET, RELU>(pacc[nb]);
        }
#pragma unroll
        for (int q = s * QPS; q < (s + 1) * QPS && q < QUADS; ++q) convert_quad(pacc, gpacc, m - 1, q);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KEEP_F32_TILE != m && (last || !SPREAD)) {      // converted right here
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) gacc[nb] = mfma_guard<ET, RELU>(acc[nb]);
    }
    if (last || !SPREAD) {
#pragma unroll
      for (int q = 0; q < QUADS; ++q) convert_quad(acc, gacc, m, q);
    }
  }
}

// 32-sample blocks per wave and workgroups per CU of the staged shading kernel (FP: the encoding layout of the instantiation).
// Two blocks per wave halve the LDS reads and the copies per MFMA.  Width 64 and 128 run two blocks at two workgroups per CU (<= 256
// registers; round 3 shipped one block at three per CU for width 128 because two blocks spilled 45 registers -- with the bias table
// in LDS and no bias registers they fit: 0.758 -> 0.709 ms at 6 x 128); width 256 runs two blocks with the whole 512-register file,
// as the 8 x 256 kernel does.  The catch-all 16-band layout parks 7 KiB of encoding per block in LDS: at width 128 two blocks would
// leave room for one workgroup only, so it keeps one block there (tuning.hpp).
template <int W, int FP = 10>
constexpr int gen_blocks() {
  return W == 64 ? 2 : W == 128 ? (FP <= 10 ? tune::kGenericBlocks128 : 1) : tune::kGenericBlocks256;
}
template <int W, int FP = 10>
constexpr int gen_occupancy() {
  return W == 64 ? (FP <= 10 ? tune::kGenericOcc64 : 2) : W == 128 ? (FP <= 10 ? tune::kGenericOcc128 : 2) : (FP <= 10 ? tune::kGenericOcc256 : 1);
}

// A5 + A6 for any shading-net topology on the 16-bit engine, weights staged per tile.  Workgroup = 4 waves x NB x 32 samples.
template <class ET, int FP, int FD, int W, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void shade_mlp16_gen_staged_kernel(ShadeArgs a, GenericTopo t) {
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD), MT = W / 32, KW = W / 16;      // KW: k-steps of a W-wide input
  constexpr int TILE = 4 * 32 * NB;
  constexpr int KSMAX = (QP / 8 + KW > KW + QD / 8) ? QP / 8 + KW : KW + QD / 8;
  constexpr int BUF = KSMAX * 1024;
  constexpr int IP = QP / 2, ID = QD / 2, IA = W / 4, IB = W / 4 + 8;      // hB also receives the (MT + 1)-tile feature (+ alpha) layer's packed part
  constexpr int STASH = (QP / 8) * 1024;      // packed position encoding of one block: [group of 4 dwords][lane], wave-private
  constexpr int BIAS_AT = 2 * BUF + 4 * NB * STASH;      // the network's bias table behind the buffers and the stashes
  // ... except where it would cost a resident workgroup: width 64 on the 16-band layout fills the CU's 160 KB with two workgroups as it is
  constexpr bool LDSB = !(W == 64 && FP > 10);
  __shared__ __attribute__((aligned(1024))) char stage_mem[BIAS_AT + (LDSB ? gen_bias_cap<W>() * 4 : 0)];
  typedef __attribute__((address_space(3))) u32x4* lds_u32x4_wptr;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  int total = a.total ? *a.total : a.max_samples;
  if (total > a.max_samples) total = a.max_samples;
  total = __builtin_amdgcn_readfirstlane(total);      // wave-uniform by construction: keeps it (and the clamp below) out of the VGPRs
  if (static_cast<int>(blockIdx.x) * TILE >= total) return;      // workgroup-uniform
  const uint32_t stash = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stage_mem)) + 2 * BUF + wave * NB * STASH + lane * 16;
  TileStage st;
  ts_start(st, a.net.w, stage_mem, wave, lane);
  ts_issue<BUF>(st, a.net.w_off[0], QP / 8, 0);
  if constexpr (LDSB) bias_table_fill<W>(reinterpret_cast<float*>(stage_mem + BIAS_AT), a.net.bias, a.net.n_bias);
  // a layer's bias blocks: LDS byte address, or (table left in global memory) byte offset from a.net.bias
  const uint32_t bl = LDSB ? static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stage_mem)) + BIAS_AT : 0u;
  auto bias_of = [&](int l) { return bl + a.net.b_off[l] * 4u; };
  const float* const gbias0 = a.net.bias;
  const float* gb = gbias0;
  float br[16];
  if constexpr (LDSB) bias_request_lds(bias_of(0), h, br);
  else bias_request(gb + a.net.b_off[0], h, br);
  const int lf = t.depth;
  for (int tile = blockIdx.x; tile * TILE < total; tile += gridDim.x) {
    if constexpr (!LDSB) gb = gbias0 + bias_launder();      // keep the bias loads inside the loops
    // narrow networks spend as long in the encodings as in their MFMAs (60 sin / cos per sample against 243 MFMAs per block at
    // 6 x 128): the position encoding is evaluated once and parked in LDS for the skip layer, the direction encoding is evaluated
    // where it is consumed -- neither lives in registers across the layer stack
    uint32_t hA[NB * IA], hB[NB * IB];
    float dpe[NB][3];
    // k-steps (= KiB) of a tile of hidden layer l (1 <= l < depth), of the feature layer at l == depth
    auto ks_of = [&](int l) { return (l < t.depth && ((t.cat_mask >> l) & 1)) ? QP / 8 + KW : KW; };
    {
      uint32_t pts[NB * IP];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float x[3];
        // past the end: the last sample again.  (The opaque zero keeps `total - 1` a per-pass scalar: as a loop invariant hipcc parked it
        // in a VGPR, spilled that, and re-loaded it from scratch at the head of every pass.)
        load_sample(a, tile * TILE + (wave * NB + nb) * 32 + j, total + bias_launder(), x, dpe[nb]);
        pe_pack<ET, FP>(x, h, pts + nb * IP);
#pragma unroll
        for (int g = 0; g < QP / 8; ++g) {
          const u32x4 v = {pts[nb * IP + 4 * g], pts[nb * IP + 4 * g + 1], pts[nb * IP + 4 * g + 2], pts[nb * IP + 4 * g + 3]};
          *((lds_u32x4_wptr)(uintptr_t)(stash + nb * STASH + g * 1024)) = v;
        }
      }
      layer_16_staged<ET, NB, BUF, QP / 8, 0, MT, true, IP, IP, IA, -1, LDSB>(st, br, a.net.w_off[0], bias_of(0), lane, pts, pts, hA, a.net.w_off[1],
                                                                              ks_of(1), bias_of(1), nullptr, gb);
    }
#pragma unroll 1
    for (int l = 1; l < t.depth; ++l) {
      if constexpr (!LDSB) gb = gbias0 + bias_launder();
      const uint32_t nxt = a.net.w_off[l + 1];
      const uint32_t nb_ = bias_of(l + 1);
      const int nks = ks_of(l + 1);
      if ((t.cat_mask >> l) & 1) {
        uint32_t pts[NB * IP];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int g = 0; g < QP / 8; ++g) {
            const u32x4 v = lds_read128(stash + nb * STASH + g * 1024);
            pts[nb * IP + 4 * g] = v[0];
            pts[nb * IP + 4 * g + 1] = v[1];
            pts[nb * IP + 4 * g + 2] = v[2];
            pts[nb * IP + 4 * g + 3] = v[3];
          }
        layer_16_staged<ET, NB, BUF, QP / 8, KW, MT, true, IP, IA, IB, -1, LDSB>(st, br, a.net.w_off[l], bias_of(l), lane, pts, hA, hB, nxt, nks, nb_, nullptr,
                                                                                 gb);      // cat([pts, h])
      } else {
        layer_16_staged<ET, NB, BUF, KW, 0, MT, true, IA, IA, IB, -1, LDSB>(st, br, a.net.w_off[l], bias_of(l), lane, hA, hA, hB, nxt, nks, nb_, nullptr, gb);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < IA; ++i) hA[nb * IA + i] = hB[nb * IB + i];
    }
    f32x16 alpha_tile[NB], rgb_tile[NB];
    layer_16_staged<ET, NB, BUF, KW, 0, MT + 1, false, IA, IA, IB, MT, LDSB>(st, br, a.net.w_off[lf], bias_of(lf), lane, hA, hA, hB, a.net.w_off[lf + 1],
                                                                             KW + QD / 8, bias_of(lf + 1), alpha_tile, gb);             // feature (+ alpha row)
    {
      uint32_t dirs[NB * ID];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) pe_pack<ET, FD>(dpe[nb], h, dirs + nb * ID);
      layer_16_staged<ET, NB, BUF, KW, QD / 8, MT / 2, true, IB, ID, IA, -1, LDSB>(st, br, a.net.w_off[lf + 1], bias_of(lf + 1), lane, hB, dirs, hA,
                                                                                    a.net.w_off[lf + 2], KW / 2, bias_of(lf + 2), nullptr, gb);      // cat([feature, dir])
    }
    const bool more = (tile + static_cast<int>(gridDim.x)) * TILE < total;      // the last tile of this pass starts the copy of the next pass's first
    layer_16_staged<ET, NB, BUF, KW / 2, 0, 1, false, IA, IA, IB, 0, LDSB>(st, br, a.net.w_off[lf + 2], bias_of(lf + 2), lane, hA, hA, hB, a.net.w_off[0],
                                                                            more ? QP / 8 : 0, bias_of(0), rgb_tile, gb);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int s = tile * TILE + (wave * NB + nb) * 32 + j;
      if (h == 0 && s < total)
        store_raw(a, s, rgb_tile[nb][0], rgb_tile[nb][1], rgb_tile[nb][2], alpha_tile[nb][0]);
    }
  }
  // no LDS-DMA may outlive the workgroup's LDS allocation.  (Nothing is in flight here -- the last pass issues no copy for a next one and
  // every tile waited for its own -- but the drain keeps that true by construction: tests/test_host_cpu.py checks it on the assembly.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- sampling network of any topology on the split-precision engine ----------------------------------------------------------------
// layer_16x3 (k_sampling16.hip.hpp) with the (hi, lo') fragment pairs fetched straight from global memory: per k-step
//   acc += Whi . xhi ;  cross += Whi . xlo' ;  cross += Wlo' . xhi         (v = acc + cross / 2048: 22-bit operands, fp32 accumulate)
// Fragments [m][s][part][lane][8 x fp16] (pack.cpp, Elem::F16_SPLIT).  LAST: fp32 outputs instead of the next layer's (hi, lo') split.
template <int KS, int MT, bool LAST>
__device__ __forceinline__ void layer_16x3_direct(const u32x4* __restrict__ w, const float* __restrict__ bias, int lane, const uint32_t* in_hi,
                                                  const uint32_t* in_lo, uint32_t* out_hi, uint32_t* out_lo, float* out_f32) {
  const int h = lane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc, cross;
    const float4* bp = reinterpret_cast<const float4*>(bias + (m * 2 + h) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = bp[g];
      acc[4 * g + 0] = b.x;
      acc[4 * g + 1] = b.y;
      acc[4 * g + 2] = b.z;
      acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cross[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 whi = w[((m * KS + s) * 2 + 0) * 64 + lane], wlo = w[((m * KS + s) * 2 + 1) * 64 + lane];
      const u32x4 bh = {in_hi[4 * s], in_hi[4 * s + 1], in_hi[4 * s + 2], in_hi[4 * s + 3]};
      const u32x4 bl = {in_lo[4 * s], in_lo[4 * s + 1], in_lo[4 * s + 2], in_lo[4 * s + 3]};
      acc = Fp16::mfma(whi, bh, acc);
      cross = Fp16::mfma(whi, bl, cross);
      cross = Fp16::mfma(wlo, bh, cross);
    }
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) epilogue_pair_16x3<LAST>(acc, cross, m, pi, out_hi, out_lo, out_f32);
  }
}

// The same layer with the (hi, lo') pairs staged per tile (2 KiB per k-step), order pinned as in layer_16_staged.
template <int BUF_BYTES, int KS, int MT, bool LAST>
__device__ __forceinline__ void layer_16x3_staged(TileStage& st, float (&br)[16], uint32_t w_off, uint32_t bias, int lane,
                                                  const uint32_t* in_hi, const uint32_t* in_lo, uint32_t* out_hi, uint32_t* out_lo, float* out_f32,
                                                  uint32_t next_off, int next_frags, uint32_t next_bias) {      // bias: LDS byte addresses
  constexpr int D = KS < 2 ? KS : 2;      // pairs requested ahead
  static_assert(KS * 2048 <= BUF_BYTES, "tile does not fit its LDS buffer");
  const int h = lane >> 5;
  // (spreading tile m - 1's pairs over tile m's k-steps out of a second accumulator pair, as layer_16_staged does at width 256, gains
  // 0.7 % at 5 x 256 -- profiles/r04_variants_generic_spread_sampling.log -- and was not kept)
  f32x16 acc, cross;      // tile m - 1's split / conversion runs behind tile m's barrier and first fragment requests (see layer_16_staged)
  auto convert = [&](int m) {
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) epilogue_pair_16x3<LAST>(acc, cross, m, pi, out_hi, out_lo, out_f32);
  };
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bool last = m == MT - 1;
    const uint32_t rd = ts_next<BUF_BYTES>(st, last ? next_off : w_off + (m + 1) * KS * 128, last ? next_frags : 2 * KS);
    u32x4 fh[D], fl[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      fh[i] = lds_read128(rd + (2 * i) * 1024);
      fl[i] = lds_read128(rd + (2 * i + 1) * 1024);
    }
    if (tune::kGenericBiasDirect && !(tune::kAblateGeneric & 2)) {
      float b16[16];
      bias_request_lds(bias + m * 128, h, b16);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = b16[r];
        cross[r] = 0.f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = br[r];
        cross[r] = 0.f;
      }
      if (tune::kAblateGeneric & 2) {      // timing ablation: no bias reads at all
#pragma unroll
        for (int r = 0; r < 16; ++r) br[r] = 0.f;
      } else {
        bias_request_lds(last ? next_bias : bias + (m + 1) * 128, h, br);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 whi = fh[s % D], wlo = fl[s % D];
      if (s + D < KS) {
        fh[s % D] = lds_read128(rd + (2 * (s + D)) * 1024);
        fl[s % D] = lds_read128(rd + (2 * (s + D) + 1) * 1024);
      }
      const u32x4 bh = {in_hi[4 * s], in_hi[4 * s + 1], in_hi[4 * s + 2], in_hi[4 * s + 3]};
      const u32x4 bl = {in_lo[4 * s], in_lo[4 * s + 1], in_lo[4 * s + 2], in_lo[4 * s + 3]};
      acc = Fp16::mfma(whi, bh, acc);
      cross = Fp16::mfma(whi, bl, cross);
      cross = Fp16::mfma(wlo, bh, cross);
      __builtin_amdgcn_sched_barrier(0);
    }
    convert(m);
  }
}

// A1 + A2 + A3 for any sampling-net topology without raySampleInput (depth 2..8, width 64 / 128 / 256) and any encoding layout.
// One wave = 32 rays, 4 waves per workgroup; writes the raw outputs (selection by select_rows_kernel, as after the fp32 generic kernel).
// STAGED: weight tiles through LDS, one copy per workgroup (tuning.hpp kGenericStaged); else every wave fetches its own from L2.
template <int FP, int FD, int W, bool STAGED>
__global__ __launch_bounds__(256, (STAGED && W <= 128 && FP <= 10) ? 2 : 1) void sample_mlp16x3_gen_kernel(SampleArgs a, GenericTopo t) {
  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP, MT = W / 32, KW = W / 16;
  constexpr int BUF = (Q0 / 8 > KW ? Q0 / 8 : KW) * 2048;
  constexpr int BIAS_AT = (STAGED ? 2 * BUF : 0) + 4 * kPairLdsBytesPerWave;      // staged: the bias table behind the selection's staging block
  __shared__ __attribute__((aligned(1024))) char stage_mem[BIAS_AT + (STAGED ? gen_bias_cap<W>() * 4 : 0)];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int blk = blockIdx.x * 4 + wave;
  if (!STAGED && blk * 32 >= a.n_rays) return;      // staged: every wave of the workgroup takes part in the copies and barriers
  const int local = blk * 32 + j;
  const bool valid = local < a.n_rays;
  const int ray = a.first_ray + (valid ? local : a.n_rays - 1);
  int col, row;
  ray_pixel(a.g, ray, &col, &row);
  float nds[3], p[3], u[3];
  gen_ray(a.g, col, row, nds, p);
  unit3(nds, u);
  uint32_t aH[W / 4], aL[W / 4], bH[W / 4], bL[W / 4];
  const u32x4* w = a.net16.w;
  const float* const bias0 = a.net16.bias;
  const float* b = bias0;
  float out[64];
  if constexpr (STAGED) {
    TileStage st;
    ts_start(st, w, stage_mem, wave, lane);
    ts_issue<BUF>(st, a.net16.w_off[0], 2 * (Q0 / 8), 0);
    bias_table_fill<W>(reinterpret_cast<float*>(stage_mem + BIAS_AT), a.net16.bias, a.net16.n_bias);
    const uint32_t bl = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stage_mem)) + BIAS_AT;
    auto bias_of = [&](int l) { return bl + a.net16.b_off[l] * 4u; };
    float br[16];
    bias_request_lds(bias_of(0), h, br);
    {
      float tt[Q0];
      uint32_t iH[Q0 / 2], iL[Q0 / 2];
      pe_eval<FD, true>(u, h, tt);            // [dir PE | pos PE]  (src/features.py:868-874)
      pe_eval<FP, true>(p, h, tt + QD);
#pragma unroll
      for (int q = 0; q < Q0 / 2; ++q) split_pack(tt[2 * q], tt[2 * q + 1], &iH[q], &iL[q]);
      layer_16x3_staged<BUF, Q0 / 8, MT, false>(st, br, a.net16.w_off[0], bias_of(0), lane, iH, iL, bH, bL, nullptr, a.net16.w_off[1], 2 * KW, bias_of(1));
    }
#pragma unroll 1
    for (int l = 1; l + 1 < t.depth; ++l) {
      layer_16x3_staged<BUF, KW, MT, false>(st, br, a.net16.w_off[l], bias_of(l), lane, bH, bL, aH, aL, nullptr, a.net16.w_off[l + 1], 2 * KW, bias_of(l + 1));
#pragma unroll
      for (int i = 0; i < W / 4; ++i) {
        bH[i] = aH[i];
        bL[i] = aL[i];
      }
    }
    const int ll = t.depth - 1;
    layer_16x3_staged<BUF, KW, 4, true>(st, br, a.net16.w_off[ll], bias_of(ll), lane, bH, bL, nullptr, nullptr, out, 0, 0, bias_of(ll));
  } else {
    {
      float tt[Q0];
      uint32_t iH[Q0 / 2], iL[Q0 / 2];
      pe_eval<FD, true>(u, h, tt);            // [dir PE | pos PE]  (src/features.py:868-874)
      pe_eval<FP, true>(p, h, tt + QD);
#pragma unroll
      for (int q = 0; q < Q0 / 2; ++q) split_pack(tt[2 * q], tt[2 * q + 1], &iH[q], &iL[q]);
      layer_16x3_direct<Q0 / 8, MT, false>(w + a.net16.w_off[0], b + a.net16.b_off[0], lane, iH, iL, bH, bL, nullptr);
    }
#pragma unroll 1
    for (int l = 1; l + 1 < t.depth; ++l) {
      asm volatile("" : "+v"(w), "+v"(b));      // keep the fragment loads inside the loop (see shade_mlp32_kernel)
      layer_16x3_direct<KW, MT, false>(w + a.net16.w_off[l], b + a.net16.b_off[l], lane, bH, bL, aH, aL, nullptr);
#pragma unroll
      for (int i = 0; i < W / 4; ++i) {
        bH[i] = aH[i];
        bL[i] = aL[i];
      }
    }
    layer_16x3_direct<KW, 4, true>(w + a.net16.w_off[t.depth - 1], b + a.net16.b_off[t.depth - 1], lane, bH, bL, nullptr, nullptr, out);
  }
  if (a.fused_select) {
    // A4 in the epilogue, as in sample_mlp16x3_kernel: the 128 raw outputs of ray j sit in lanes j and j + 32 (k_select_pair.hip.hpp)
    const uint32_t sel_stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stage_mem)) + (STAGED ? 2 * BUF : 0) + wave * kPairLdsBytesPerWave + lane * 16;
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) z = __builtin_fmaf(out[i], 0.f, z);    // NaN iff some output is inf / NaN (fp16 range left)
    const bool bad_ray = (z != z) | (pair_xchg(static_cast<uint32_t>(z != z)) != 0u);
    if (bad_ray && valid && h == 0 && a.overflow_flag) atomicAdd(a.overflow_flag, 1);
    pair_epilogue<false>(out, lane, local, valid, sel_stage, a.sel);      // never pass 2 of the guarded selection
  }
  if (valid) {
    if (a.oracle_out) {
      float* o = a.oracle_out + static_cast<size_t>(local) * kBins;
      bool bad = false;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(out[16 * m + 4 * g], out[16 * m + 4 * g + 1], out[16 * m + 4 * g + 2], out[16 * m + 4 * g + 3]);
          bad |= !(fabsf(v.x) < 3.0e38f) | !(fabsf(v.y) < 3.0e38f) | !(fabsf(v.z) < 3.0e38f) | !(fabsf(v.w) < 3.0e38f);
          *reinterpret_cast<float4*>(o + 32 * m + 8 * g + 4 * h) = v;
        }
      const bool bad_ray = bad | (pair_xchg(static_cast<uint32_t>(bad)) != 0u);      // an activation left the fp16 range
      if (bad_ray && h == 0 && a.overflow_flag) atomicAdd(a.overflow_flag, 1);
    }
    if (a.rays_out) {
      float ro[3] = {p[0], p[1], p[2]}, rd[3] = {nds[0], nds[1], nds[2]};
      if (a.g.use_ndc) ndc_ray(a.g, p, nds, ro, rd);
      float4* r = reinterpret_cast<float4*>(a.rays_out + static_cast<size_t>(local) * 8);
      if (h == 0) r[0] = make_float4(ro[0], ro[1], ro[2], 0.f);
      else r[1] = make_float4(rd[0], rd[1], rd[2], 0.f);
    }
  }
  if constexpr (STAGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup's LDS allocation (see above)
}

}  // namespace adanerf
