// tile_dispatch.hpp -- the (LR, LC) tile shapes instantiated for every (type, pass mode), and the
// runtime switch over them.  Included by tile_<type>_<mode>.hip so the four files compile in parallel.
#pragma once

#include "plan.hpp"
#include "tile_fft.hpp"


namespace phast {

template <typename T, bool PRE_TW, bool TRANSPOSE>
hipError_t launch_tile_mode(int lr, int lc, bool plane_seq, unsigned grid, hipStream_t stream, const TileArgs &a,
                            bool query_only, int *blocks_per_cu, size_t *lds, hipEvent_t e0, hipEvent_t e1) {
    // plane-sequential LDS exchange: f64 default (half the LDS -> more workgroups per CU).  The two-plane form
    // (half the barriers) exists for every f32 shape and for the 4096-point f64 shapes (latency plans).
    if (sizeof(T) == 8 && !(lr + lc == 12)) plane_seq = true;
    if (sizeof(T) == 4) plane_seq = false;
#define PHAST_CASE(LR_, LC_)                                                                                        \
    if (lr == LR_ && lc == LC_) {                                                                                   \
        if constexpr (sizeof(T) == 8) {                                                                             \
            if (plane_seq)                                                                                          \
                return launch_tile_inst<T, LR_, LC_, PRE_TW, TRANSPOSE, true>(grid, stream, a, query_only,          \
                                                                              blocks_per_cu, lds, e0, e1);          \
            if constexpr (LR_ + LC_ == 12)                                                                          \
                return launch_tile_inst<T, LR_, LC_, PRE_TW, TRANSPOSE, false>(grid, stream, a, query_only,         \
                                                                               blocks_per_cu, lds, e0, e1);         \
        } else {                                                                                                    \
            return launch_tile_inst<T, LR_, LC_, PRE_TW, TRANSPOSE, false>(grid, stream, a, query_only,             \
                                                                           blocks_per_cu, lds, e0, e1);             \
        }                                                                                                           \
    }
    PHAST_TILE_SHAPES(PHAST_CASE)
#undef PHAST_CASE
    return hipErrorInvalidValue;
}

// defined in tile_f64_a.hip, tile_f64_bc.hip, tile_f32_a.hip, tile_f32_bc.hip
hipError_t launch_tile_f64_a(int lr, int lc, bool plane_seq, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_tile_f64_bc(int lr, int lc, bool plane_seq, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_tile_f32_a(int lr, int lc, bool plane_seq, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_tile_f32_bc(int lr, int lc, bool plane_seq, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

}  // namespace phast
