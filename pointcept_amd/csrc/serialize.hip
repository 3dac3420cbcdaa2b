// serialize.hip -- space-filling-curve keys (Z-order / Hilbert, optional x<->y swap, batch prefix)
// for all requested orders in ONE pass over the coordinates.
//
// Replaces pointcept/models/utils/serialization/default.py:8-24 (dispatch),
//          .../z_order.py:66-101 (LUT Morton interleave: x bit i -> 3i+2, y -> 3i+1, z -> 3i),
//          .../hilbert.py:91-192 (Skilling transform on bit planes + Gray->binary prefix XOR).
// The reference runs ~depth*3*6 elementwise launches per Hilbert order; this is one HBM-bound
// kernel: read 3 coords + batch, write k int64 codes per point (fully coalesced per code row).
#include "ptc_common.h"
#include "sfc_keys.h"

#define PTC_MAX_ORDERS 8
struct OrderList { int n; int o[PTC_MAX_ORDERS]; };

template <typename CoordT>
__global__ void __launch_bounds__(256)
serialize_encode_kernel(const CoordT* __restrict__ gc, const int64_t* __restrict__ batch, int64_t n,
                        int depth, OrderList orders, int64_t* __restrict__ code_out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint32_t mask = (depth >= 32) ? 0xffffffffu : ((1u << depth) - 1u);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t x = (uint32_t)gc[3 * i + 0] & mask;
    const uint32_t y = (uint32_t)gc[3 * i + 1] & mask;
    const uint32_t z = (uint32_t)gc[3 * i + 2] & mask;
    const uint64_t prefix = batch ? ((uint64_t)batch[i] << (3 * depth)) : 0ull;
#pragma unroll
    for (int r = 0; r < PTC_MAX_ORDERS; ++r) {
      if (r < orders.n) {
        uint64_t key;
        switch (orders.o[r]) {
          case PTC_ORDER_Z: key = ptc_morton3(x, y, z); break;
          case PTC_ORDER_Z_TRANS: key = ptc_morton3(y, x, z); break;        // default.py:14
          case PTC_ORDER_HILBERT: key = ptc_hilbert3(x, y, z, depth); break;
          default: key = ptc_hilbert3(y, x, z, depth); break;               // default.py:18
        }
        code_out[(int64_t)r * n + i] = (int64_t)(prefix | key);             // default.py:21-23
      }
    }
  }
}

extern "C" int ptc_serialize_encode(const void* grid_coord, int coord_is_i64, const int64_t* batch,
                                    int64_t n, int depth, const int* orders, int k,
                                    int64_t* code_out, ptc_stream_t stream) {
  PTC_REQUIRE(n >= 0, PTC_EINVAL, "ptc_serialize_encode: n < 0");
  PTC_REQUIRE(depth >= 1 && depth <= 16, PTC_EINVAL, "ptc_serialize_encode: depth %d not in [1,16]", depth);
  PTC_REQUIRE(k >= 1 && k <= PTC_MAX_ORDERS, PTC_EINVAL, "ptc_serialize_encode: k=%d not in [1,%d]", k, PTC_MAX_ORDERS);
  PTC_REQUIRE(orders != nullptr, PTC_EINVAL, "ptc_serialize_encode: orders is null");
  OrderList ol;
  ol.n = k;
  for (int i = 0; i < PTC_MAX_ORDERS; ++i) ol.o[i] = 0;
  for (int i = 0; i < k; ++i) {
    PTC_REQUIRE(orders[i] >= 0 && orders[i] <= 3, PTC_EINVAL, "ptc_serialize_encode: bad order %d", orders[i]);
    ol.o[i] = orders[i];
  }
  if (n == 0) return PTC_OK;
  PTC_REQUIRE(grid_coord && code_out, PTC_EINVAL, "ptc_serialize_encode: null buffer");
  const int block = 256;
  int64_t grid = ptc_cdiv(n, block);
  if (grid > 256 * 16) grid = 256 * 16;  // grid-stride beyond 16 blocks per CU
  hipStream_t s = (hipStream_t)stream;
  if (coord_is_i64)
    hipLaunchKernelGGL(serialize_encode_kernel<int64_t>, dim3((unsigned)grid), dim3(block), 0, s,
                       (const int64_t*)grid_coord, batch, n, depth, ol, code_out);
  else
    hipLaunchKernelGGL(serialize_encode_kernel<int32_t>, dim3((unsigned)grid), dim3(block), 0, s,
                       (const int32_t*)grid_coord, batch, n, depth, ol, code_out);
  PTC_CHECK_LAUNCH("serialize_encode_kernel");
  return PTC_OK;
}
