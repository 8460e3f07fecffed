// shapes.hpp — query shapes of the LDS fold kernels (fast_path.hip, leaf_path.hip): the fold code is written once against a
// "shape" (value dtypes + accumulator list).  RtShape reads it from the Plan at run time (any query); StShape carries it in
// its type, so the accumulator loop unrolls and every dtype / op switch folds away.
#pragma once
#include <stdint.h>

#include "device.hpp"
#include "plan.hpp"

namespace {
// The fold code is written once against a "shape" (value dtypes + accumulator list).  RtShape reads it from
// the Plan at run time (any query); StShape carries it in its type, so the accumulator loop unrolls and every
// dtype / op switch folds away: straight-line code for the shapes instantiated at the end of this file.
#define ACC_ENC(op, conv, src, width) ((uint32_t)(op) | ((uint32_t)(conv) << 8) | ((uint32_t)((src) + 1) << 16) | ((uint32_t)(width) << 24))
template <uint32_t... V>
struct USeq {
  static constexpr int N = (int)sizeof...(V);
  __host__ __device__ __forceinline__ static constexpr uint32_t at(int i) {
    constexpr uint32_t a[N + 1] = {V..., 0u};
    return a[i];
  }
};
struct RtShape {
  static constexpr bool STATIC = false;
  static constexpr int NA = 0;
  DEVINL static int n_vals(const Plan &P) { return P.n_vals; }
  DEVINL static uint32_t val_dtype(const Plan &P, int c) { return P.vals[c].dtype; }
  DEVINL static bool val_nullable(const Plan &P, int c) { return P.vals[c].validity != nullptr; }
  DEVINL static int n_accs(const Plan &P) { return P.n_accs; }
  DEVINL static uint32_t acc_op(const Plan &P, int a) { return P.accs[a].op; }
  DEVINL static uint32_t acc_conv(const Plan &P, int a) { return P.accs[a].conv; }
  DEVINL static int acc_src(const Plan &P, int a) { return P.accs[a].src; }
  DEVINL static uint32_t acc_width(const Plan &P, int a) { return P.accs[a].width; }
};
template <typename VD, typename AC> // value dtypes (no validity bitmaps), ACC_ENC-coded accumulators
struct StShape {
  static constexpr bool STATIC = true;
  static constexpr int NA = AC::N;
  __host__ __device__ __forceinline__ static constexpr int n_vals(const Plan &) { return VD::N; }
  __host__ __device__ __forceinline__ static constexpr uint32_t val_dtype(const Plan &, int c) { return VD::at(c); }
  __host__ __device__ __forceinline__ static constexpr bool val_nullable(const Plan &, int) { return false; }
  __host__ __device__ __forceinline__ static constexpr int n_accs(const Plan &) { return AC::N; }
  __host__ __device__ __forceinline__ static constexpr uint32_t acc_op(const Plan &, int a) { return AC::at(a) & 0xFFu; }
  __host__ __device__ __forceinline__ static constexpr uint32_t acc_conv(const Plan &, int a) { return (AC::at(a) >> 8) & 0xFFu; }
  __host__ __device__ __forceinline__ static constexpr int acc_src(const Plan &, int a) { return (int)((AC::at(a) >> 16) & 0xFFu) - 1; }
  __host__ __device__ __forceinline__ static constexpr uint32_t acc_width(const Plan &, int a) { return AC::at(a) >> 24; }
};

template <int I, int N, typename F>
DEVINL void static_for(F &&f) {
  if constexpr (I < N) {
    f(I);
    static_for<I + 1, N>(f);
  }
}


// ---- pre-instantiated shapes (accumulators in lowered order: sorted by source column, counters first) ------
using ShSumI32 = StShape<USeq<PLHIP_I32>, USeq<ACC_ENC(ACC_ADD_U64, CONV_INT, 0, 8)>>;                                  // q1
using ShSumI32MeanF64 = StShape<USeq<PLHIP_I32, PLHIP_F64>,                                                             // q3
                                USeq<ACC_ENC(ACC_CNT_ALL, CONV_NONE, -1, 4), ACC_ENC(ACC_ADD_U64, CONV_INT, 0, 8),
                                     ACC_ENC(ACC_ADD_F64, CONV_F64, 1, 8)>>;
using ShSum3 = StShape<USeq<PLHIP_I32, PLHIP_I32, PLHIP_F64>,                                                           // q5
                       USeq<ACC_ENC(ACC_ADD_U64, CONV_INT, 0, 8), ACC_ENC(ACC_ADD_U64, CONV_INT, 1, 8),
                            ACC_ENC(ACC_ADD_F64, CONV_F64, 2, 8)>>;
using ShWide = StShape<USeq<PLHIP_I64, PLHIP_F64>,                                         // i64 sum/min/max, f64 sum/mean, count
                       USeq<ACC_ENC(ACC_CNT_ALL, CONV_NONE, -1, 4), ACC_ENC(ACC_ADD_U64, CONV_INT, 0, 8),
                            ACC_ENC(ACC_MIN_I64, CONV_INT, 0, 8), ACC_ENC(ACC_MAX_I64, CONV_INT, 0, 8),
                            ACC_ENC(ACC_ADD_F64, CONV_F64, 1, 8)>>;
using ShSumF64 = StShape<USeq<PLHIP_F64>, USeq<ACC_ENC(ACC_ADD_F64, CONV_F64, 0, 8)>>;
using ShSumI64 = StShape<USeq<PLHIP_I64>, USeq<ACC_ENC(ACC_ADD_U64, CONV_INT, 0, 8)>>;
using ShWideF = StShape<USeq<PLHIP_F64, PLHIP_I64>,                                        // the same with the f64 column first (config 5)
                        USeq<ACC_ENC(ACC_CNT_ALL, CONV_NONE, -1, 4), ACC_ENC(ACC_ADD_F64, CONV_F64, 0, 8),
                             ACC_ENC(ACC_ADD_U64, CONV_INT, 1, 8), ACC_ENC(ACC_MIN_I64, CONV_INT, 1, 8),
                             ACC_ENC(ACC_MAX_I64, CONV_INT, 1, 8)>>;
#define FP_SHAPES(X) X(0, ShSumI32) X(1, ShSumI32MeanF64) X(2, ShSum3) X(3, ShWide) X(4, ShSumF64) X(5, ShSumI64) X(6, ShWideF)

template <typename SH>
bool shape_matches(const Plan &p) {
  if (p.n_vals != SH::n_vals(p) || p.n_accs != SH::n_accs(p)) return false;
  for (int c = 0; c < p.n_vals; c++)
    if (p.vals[c].dtype != SH::val_dtype(p, c) || p.vals[c].validity) return false;
  for (int a = 0; a < p.n_accs; a++)
    if (p.accs[a].op != SH::acc_op(p, a) || p.accs[a].conv != SH::acc_conv(p, a) || p.accs[a].src != SH::acc_src(p, a) ||
        p.accs[a].width != SH::acc_width(p, a))
      return false;
  return true;
}
int find_shape(const Plan &p) {
#define X(i, SH) if (shape_matches<SH>(p)) return i;
  FP_SHAPES(X)
#undef X
  return -1;
}

} // namespace
