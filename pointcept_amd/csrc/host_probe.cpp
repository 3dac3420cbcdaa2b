// host_probe.cpp -- g++ build of the pure helper functions the gfx950 kernels are made of
// (sfc_keys.h, pad_maps.h, voxel_hash.h, voxel_keys.h), exported for CPU unit checks (-m "not gpu" tests).
// This is PRODUCT code exercised on the host, not an oracle: the same headers are compiled into
// libptcore.so by hipcc.  The oracle they are compared against lives in oracle/.
#include <stdint.h>
#include "sfc_keys.h"
#include "pad_maps.h"
#include "voxel_hash.h"
#include "voxel_keys.h"

extern "C" {

// mirrors serialize_encode_kernel's per-point body
void probe_serialize_encode(const int64_t* gc, const int64_t* batch, int64_t n, int depth, const int* orders, int k,
                            int64_t* code_out) {
  const uint32_t mask = (1u << depth) - 1u;
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t x = (uint32_t)gc[3 * i] & mask, y = (uint32_t)gc[3 * i + 1] & mask, z = (uint32_t)gc[3 * i + 2] & mask;
    const uint64_t prefix = batch ? ((uint64_t)batch[i] << (3 * depth)) : 0ull;
    for (int r = 0; r < k; ++r) {
      uint64_t key;
      switch (orders[r]) {
        case 0: key = ptc_morton3(x, y, z); break;
        case 1: key = ptc_morton3(y, x, z); break;
        case 2: key = ptc_hilbert3(x, y, z, depth); break;
        default: key = ptc_hilbert3(y, x, z, depth); break;
      }
      code_out[(int64_t)r * n + i] = (int64_t)(prefix | key);
    }
  }
}

// mirrors patch_pad_maps_kernel (serial over positions)
void probe_patch_pad_maps(const int64_t* offset, int B, int64_t K, int64_t n, int64_t n_pad, int64_t n_seq,
                          int64_t* pad, int64_t* unpad, int32_t* cu_seqlens, int64_t* dup) {
  int64_t* s_off = new int64_t[B];
  int64_t* s_offpad = new int64_t[B];
  int64_t* s_seq = new int64_t[B];
  int64_t prev = 0, accp = 0, accs = 0;
  for (int i = 0; i < B; ++i) {
    s_off[i] = offset[i];
    const int64_t ni = s_off[i] - prev;
    prev = s_off[i];
    accp += ptc_padded_len(ni, K);
    accs += ptc_num_seq(ni, K);
    s_offpad[i] = accp;
    s_seq[i] = accs;
  }
  for (int64_t t = 0; t < n_pad; ++t) {
    const int i = ptc_find_scene(s_offpad, B, t);
    const int64_t o0 = i ? s_off[i - 1] : 0, p0 = i ? s_offpad[i - 1] : 0;
    pad[t] = o0 + ptc_pad_local(t - p0, s_off[i] - o0, K);
  }
  for (int64_t t = 0; t < n; ++t) {
    const int i = ptc_find_scene(s_off, B, t);
    const int64_t o0 = i ? s_off[i - 1] : 0, p0 = i ? s_offpad[i - 1] : 0;
    unpad[t] = t - o0 + p0;
    if (dup) {
      const int64_t d = ptc_dup_local(t - o0, s_off[i] - o0, K);
      dup[t] = d < 0 ? -1 : p0 + d;
    }
  }
  for (int64_t t = 0; t <= n_seq; ++t) {
    if (t < n_seq) {
      const int i = ptc_find_scene(s_seq, B, t);
      const int64_t q0 = i ? s_seq[i - 1] : 0, p0 = i ? s_offpad[i - 1] : 0;
      cu_seqlens[t] = (int32_t)(p0 + (t - q0) * K);
    } else {
      cu_seqlens[t] = (int32_t)n_pad;
    }
  }
  delete[] s_off;
  delete[] s_offpad;
  delete[] s_seq;
}

int64_t probe_padded_len(int64_t n_i, int64_t K) { return ptc_padded_len(n_i, K); }
int64_t probe_num_seq(int64_t n_i, int64_t K) { return ptc_num_seq(n_i, K); }
uint64_t probe_vox_pack(int b, int x, int y, int z) { return ptc_vox_pack(b, x, y, z); }
uint64_t probe_vox_hash(uint64_t h) { return ptc_vox_hash(h); }


// mirrors voxel_floor_kernel + voxel_key_kernel (voxelize.hip)
void probe_voxel_keys(const float* coord, int64_t n, double grid_size, int64_t* grid, int64_t* min3, int64_t* key) {
  long long m[3] = {0x7fffffffffffffffll, 0x7fffffffffffffffll, 0x7fffffffffffffffll};
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const long long v = ptc_voxel_floor(coord[3 * i + a], grid_size);
      grid[3 * i + a] = v;
      if (v < m[a]) m[a] = v;
    }
  for (int a = 0; a < 3; ++a) min3[a] = n ? m[a] : 0;
  for (int64_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) grid[3 * i + a] -= m[a];
    key[i] = (int64_t)ptc_fnv3((unsigned long long)grid[3 * i], (unsigned long long)grid[3 * i + 1], (unsigned long long)grid[3 * i + 2]);
  }
}

// Lovasz steps of one class row given its sorted foreground flags (mirrors lovasz_step_kernel's arithmetic)
void probe_lovasz_steps(const int32_t* fg_sorted, int64_t n, double* step) {
  long long gts = 0, cf = 0;
  for (int64_t i = 0; i < n; ++i) gts += fg_sorted[i];
  for (int64_t i = 0; i < n; ++i) {
    cf += fg_sorted[i];
    step[i] = gts > 0 ? ptc_lovasz_step(gts, cf, (i + 1) - cf, fg_sorted[i]) : 0.0;
  }
}
}
