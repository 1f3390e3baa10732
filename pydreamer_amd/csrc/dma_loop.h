// The LDS-DMA operand pipeline shared by gemm_dma_kernel (gemm.hip) and panel32_kernel (panel.hip): per-lane source addressing of
// global_load_lds_dwordx4 pieces (DmaOperand), the fragment registers of one 8-k group (DmaFrag) and the hand-placed waits.
// See the comment above gemm_dma_kernel for the design.  gfx950 only.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
// a ragged last k-tile reads its k-groups past the end from here (16 bytes: one LDS-DMA chunk)
__device__ __attribute__((aligned(16))) const float dm_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ void dma_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void dma_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ROWS, int LAYOUT, bool GATHER>
struct DmaOperand {
  static constexpr int NP = ROWS / 32;        // 1-KiB pieces of a (ROWS x 32) tile per wave (4 waves)
  const float* P;
  const int* tk;                              // GATHER: the table indexed by k (layout 0: minor table, layout 1: major table)
  int ld;
  const float* base[GATHER ? 1 : NP];         // plain: address of this lane's 16-byte chunk in the tile at k = 0 (a k-step adds a UNIFORM
                                              //   offset: no per-lane multiply, no branch around one - hipcc branches around 64-bit products)
  unsigned off[GATHER ? NP : 1];              // GATHER: k-independent element offset (the other table's entry)
  int kof[NP];                                // the chunk's k offset inside a 32-k tile
  int tv[GATHER ? NP : 1];                    // GATHER: tk[k] of the tile to issue next (fetched one tile ahead)
  __device__ __forceinline__ void init(const float* P_, int ld_, int row0, int nrows, const int* tmaj, const int* tmin, int wave,
                                       int lane) {
    P = P_; ld = ld_;
    tk = LAYOUT == 0 ? tmin : tmaj;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = wave + 4 * i;
      if (LAYOUT == 0) {                      // piece = 8 rows x 128 B; lane -> (row l >> 3, LDS slot l & 7); fetches chunk slot ^ swz(row)
        const int row = 8 * p + (lane >> 3);
        kof[i] = (((lane & 7) ^ ((row >> 1) & 7)) << 2);
        const int gr = min(row0 + row, nrows - 1);
        if constexpr (GATHER) off[i] = (unsigned)tmaj[gr];
        else base[i] = P + (size_t)gr * ld + kof[i];
      } else {                                // image [k][ROWS]: 16-byte chunk e = 64 p + lane -> (k e / (ROWS/4), rows 4 (e % (ROWS/4)) ..)
        constexpr int CPR = ROWS / 4;
        const int e = p * 64 + lane;
        kof[i] = e / CPR;
        const int col = min(row0 + ((e % CPR) << 2), nrows - 4);
        if constexpr (GATHER) off[i] = (unsigned)tmin[col];
        else base[i] = P + (size_t)kof[i] * ld + col;
      }
    }
    if constexpr (GATHER)
#pragma unroll
      for (int i = 0; i < NP; ++i) tv[i] = 0;
  }
  __device__ __forceinline__ void fetch_tab(int k0, int kend) {
    if constexpr (GATHER) {
#pragma unroll
      for (int i = 0; i < NP; ++i) tv[i] = tk[min(k0 + kof[i], kend - 1)];
    }
  }
  __device__ __forceinline__ void issue(int k0, int kend, unsigned char* img, int wave) const {
    const size_t koff = LAYOUT == 0 ? (size_t)k0 : (size_t)k0 * (size_t)ld;      // uniform
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool ok = k0 + kof[i] < kend;
      const float* at;
      if constexpr (GATHER) at = P + ((size_t)off[i] + (size_t)tv[i]);
      else at = base[i] + koff;
      const uintptr_t src = ok ? (uintptr_t)at : (uintptr_t)dm_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(img + (wave + 4 * i) * 1024), 16, 0, 0);
    }
  }
};

// One 8-k group of fragments of one 32-row block: 4 floats per lane (k = 8 kg + 4 half + j).
template <int LAYOUT> struct DmaFrag;
template <> struct DmaFrag<0> {
  f32x4v v;
  __device__ __forceinline__ float get(int j) const { return v[j]; }
  static constexpr int READS = 1;
};
template <> struct DmaFrag<1> {
  f32x2v lo, hi;
  __device__ __forceinline__ float get(int j) const { return j < 2 ? lo[j & 1] : hi[j & 1]; }
  static constexpr int READS = 2;
};

