// gemm_epilogue.h — operand traits (Mma<T>) and helpers of the 256x256 kernel (gemm256.hip; also used by the experimental
// kernels under tools/experiments/).  The accumulators come out of v_mfma_f32_32x32x*
// with SWAPPED operands (D = W_frag · A_frag^T):
//   acc[it][j][rq*4+e]: row = m_w + it*32 + (lane & 31) ; col = n_w + j*32 + rq*8 + (lane >> 5)*4 + e
// so a lane owns 4 consecutive output columns of one row, and every store path below first transposes through the
// wave's private 8-KiB LDS scratch `ep` so that global stores are 16 B per lane and 128-256 B contiguous per row.
// Epilogues: bias / weight scale / folded LayerNorm, activation, then one of: 16-bit rows (EPI_F16), fp8 rows
// (EPI_F8), f32 rows + residual (+ 16-bit copy + LayerNorm row partials) (EPI_F32), patch rows (EPI_PATCH), the
// per-head Q / K / V scatter in all its layouts (EPI_HEADS).
#pragma once
#include "common.h"

// sum over the 16 lanes of a DPP row (all 16 end up with the total): rotations by 8, 4, 2, 1
template <int ROR>
__device__ __forceinline__ float row_ror_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROR, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = row_ror_add<8>(v);
  v = row_ror_add<4>(v);
  v = row_ror_add<2>(v);
  return row_ror_add<1>(v);
}

// A raw buffer descriptor (stride 0: byte offsets, range-checked against `bytes`) over [base, base + bytes).  The base
// is passed through v_readfirstlane: it is wave-uniform by construction wherever this is used, but under the SGPR
// pressure of the big GEMM kernels the compiler otherwise leaves descriptor words in VGPRs and wraps every buffer
// instruction in a "waterfall" loop (readfirstlane x 4, compare, exec-mask — a dozen instructions per access).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}

// 16 x 16 REDUCE-SCATTER over a DPP row: every lane brings 16 values; lane i of each 16 leaves with value i summed over
// the 16 lanes.  Four halving stages — a lane keeps the half of its values that its partner's side of the row does not
// own, and adds the partner's copy of them — over the pairings row_mirror (i <-> 15 - i: splits on bit 3),
// row_half_mirror (i <-> 7 - i within 8: bit 2), quad_perm [2,3,0,1] (bit 1), quad_perm [1,0,3,2] (bit 0): 15 DPP adds
// and 30 selects, against 4 DPP moves + 4 adds for EACH of the 16 values as separate all-reduces.  The summation order is
// fixed (a property of this function), so every kernel that uses it produces the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_reduce_scatter(const float (&v)[16], int lane) {
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
  float w8[8], w4[4], w2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) w8[k] = (b3 ? v[k + 8] : v[k]) + dpp_f32<0x140>(b3 ? v[k] : v[k + 8]);
#pragma unroll
  for (int k = 0; k < 4; ++k) w4[k] = (b2 ? w8[k + 4] : w8[k]) + dpp_f32<0x141>(b2 ? w8[k] : w8[k + 4]);
#pragma unroll
  for (int k = 0; k < 2; ++k) w2[k] = (b1 ? w4[k + 2] : w4[k]) + dpp_f32<0x4E>(b1 ? w4[k] : w4[k + 2]);
  return (b0 ? w2[1] : w2[0]) + dpp_f32<0xB1>(b0 ? w2[0] : w2[1]);
}

// What the main loop needs to know about the operand type: a 128-byte LDS row holds one K-tile (64 x 16-bit or
// 128 x fp8), consumed in KS MFMA k-steps; a fragment is the lane's share of one k-step of one 32-row tile.
template <typename T>
struct Mma {   // f16 / bf16: v_mfma_f32_32x32x16, 16 k per step, one 16-byte slot per lane
  static constexpr int KS = 4;
  typedef typename Elt<T>::x8 Frag;
  static __device__ __forceinline__ Frag load(const char* row, int ks, int hi, int sw) {
    return *(const Frag*)(row + (((ks * 2 + hi) ^ sw) << 4));
  }
  static __device__ __forceinline__ f32x16 mma(Frag w, Frag a, f32x16 c) { return Elt<T>::mfma32(w, a, c); }
};
template <>
struct Mma<fp8> {   // e4m3: v_mfma_scale_f32_32x32x64_f8f6f4 (scales 2^0), 64 k per step, two 16-byte slots per lane
  static constexpr int KS = 2;
  typedef i32x8 Frag;
  static __device__ __forceinline__ Frag load(const char* row, int ks, int hi, int sw) {
    // operand layout of the 32x32x64 instruction (tools/micro/mx_fp8_layout.hip, measured): lane l holds row l % 32 and
    // the 32 consecutive k of half l / 32 — two adjacent 16-byte slots of the 128-byte row
#ifdef VIDIL_FP8_LAYOUT_INTERLEAVED   // alternative layout (16-byte halves interleaved), kept for the probe
    const i32x4 lo = *(const i32x4*)(row + (((ks * 4 + hi) ^ sw) << 4));
    const i32x4 hi4 = *(const i32x4*)(row + (((ks * 4 + 2 + hi) ^ sw) << 4));
#else
    const i32x4 lo = *(const i32x4*)(row + (((ks * 4 + hi * 2) ^ sw) << 4));
    const i32x4 hi4 = *(const i32x4*)(row + (((ks * 4 + hi * 2 + 1) ^ sw) << 4));
#endif
    return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);   // (a register-tuple concat, no copies)
  }
  static __device__ __forceinline__ f32x16 mma(Frag w, Frag a, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, a, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
};

// T: operand type of A / W (only its size matters here: fp8 weights carry a per-column scale); TO: 16-bit output type.
// st_s / st_ss (FOLD only): per row tile `it`, rstd and mean * rstd of the lane's row.
