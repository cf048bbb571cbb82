// gfx950 (CDNA4, wave64) kernels of the AdaNeRF per-frame hot path.  Device code only; the C ABI
// in adanerf_hip.hip launches these.  Stage map:
//   A1+A2+A3  sample_mlp_kernel      ray gen -> sphere exit -> oracle PE -> 8-layer sampling MLP (fp32 MFMA)
//   A4        select_kernel / scan_blocks_kernel / expand_kernel   top-N + threshold, deterministic compaction
//   A5+A6     shade_mlp16_kernel / shade_mlp32_kernel   fused PE + 8x256 shading MLP (bf16/f16/f32 MFMA)
//   A7        composite_kernel       sigmoid + alpha * oracle weight, front-to-back
// plus explicit-feature debug kernels (ray_features_kernel, shade_features_kernel) that materialise
// what the reference launchers wrote to memory, for parity tests only.
// The code lives in one header per stage; this file includes them all.
#pragma once
#include "k_common.hip.hpp"
#include "k_mlp_f32.hip.hpp"
#include "k_compact.hip.hpp"
#include "k_select_pair.hip.hpp"
#include "k_mlp16.hip.hpp"
#include "k_sampling16.hip.hpp"
#include "k_donerf.hip.hpp"
#include "k_coarse_fine.hip.hpp"
#include "k_composite.hip.hpp"
