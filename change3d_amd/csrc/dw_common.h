// Pieces shared by the depthwise 3x3x3 translation units (dw_conv.hip, dw_bwd_fused.hip).
#pragma once
#include "common.h"

namespace {

constexpr int DW_CV = 4;    // channel vectors (of 8) per workgroup pass = 32 channels
constexpr int DW_MAXT = 5;

struct DwGeom {
  int B, T, H, W, Ho, Wo, C, Cp, stride;
};

// XCD-aware workgroup order.  A 32-channel chunk reads 64 B of every 112/224/432-byte pixel row, so
// the chunks of one tile share their 128-byte lines: counters showed the depthwise kernels fetching
// ~1.8x their input when sibling chunks ran far apart (3-D grid: chunk = blockIdx.y).  Workgroups are
// dealt round-robin to the 8 XCDs (one L2 each) in flattened-id order, so a 1-D grid decoded as
//   id -> (xcd = id % 8, k = id / 8), chunk = k % chunks, group = (k / chunks) * 8 + xcd
// makes the chunks of a group consecutive arrivals on ONE XCD: the second chunk hits that L2.
constexpr int N_XCD = 8;
struct ChunkOrder {
  int chunk, group;   // group = index over (tile groups x samples); < 0: padding workgroup
};
__device__ __forceinline__ ChunkOrder chunk_order(const int chunks, const int ngroups) {
  const int id = blockIdx.x, xcd = id % N_XCD, k = id / N_XCD;
  ChunkOrder o;
  o.chunk = k % chunks;
  o.group = (k / chunks) * N_XCD + xcd;
  if (o.group >= ngroups) o.group = -1;
  return o;
}
inline unsigned chunk_order_grid(const int chunks, const long ngroups) {
  return (unsigned)(((ngroups + N_XCD - 1) / N_XCD) * N_XCD * chunks);
}

inline bool geom_ok(const DwGeom& g) {
  if (g.B <= 0 || g.T <= 0 || g.T > DW_MAXT || g.H <= 0 || g.W <= 0 || g.C <= 0 || g.Cp < g.C || (g.Cp & 7))
    return false;
  if (g.stride != 1 && g.stride != 2) return false;
  if (g.Ho != (g.H - 1) / g.stride + 1 || g.Wo != (g.W - 1) / g.stride + 1) return false;
  return true;
}

}  // namespace
