// tile_dispatch.hpp -- the (LR, LC) tile shapes instantiated for every (type, pass mode), and the
// runtime switch over them.  Included by tile_<type>_<mode>.hip so the four files compile in parallel.
#pragma once

#include "plan.hpp"
#include "tile_fft.hpp"


namespace phast {

// The 1024 x 16 f64 pre-twiddle pass at 32 points per thread (second pass of the batched 2^20 transforms, BASELINE
// configs[4]) sits exactly at the 256-register budget of a 512-thread workgroup: under the default scheduling strategy
// the compiler spills 10 VGPRs (44 bytes of scratch per lane) in the second radix-32 step, under
// -amdgpu-sched-strategy=max-ilp it does not (and the pass runs 1.5 % faster; the other kernels do not gain from that
// strategy, profiles/r03_sched_strategy_max_ilp_cmp.log).  The strategy is a per-translation-unit option, so this one
// instantiation lives in tile_f64_bc_wide.hip (phastft_amd/build.py: UNIT_FLAGS) and is only declared here.
extern template hipError_t launch_tile_inst<double, 10, 4, 5, true, false, true>(unsigned, hipStream_t, const TileArgs &, bool, int *,
                                                                                 size_t *, hipEvent_t, hipEvent_t);
// Its f32 twin, the 1024 x 32 pass on 1024 threads (128 registers per lane): 19 spilled VGPRs under the default
// options -- the SLP vectoriser pairs f32 operations into v_pk_add/v_pk_fma at the price of ~180 extra v_mov and
// even-aligned register pairs -- none with -fno-slp-vectorize and the max-ilp strategy (tile_f32_bc_wide.hip).
extern template hipError_t launch_tile_inst<float, 10, 5, 5, true, false, true>(unsigned, hipStream_t, const TileArgs &, bool, int *,
                                                                                size_t *, hipEvent_t, hipEvent_t);

// LDS exchange flavour per instantiation: f64 tiles with 16 points per thread exchange the re and im planes one
// after the other (half the LDS -> more workgroups per CU); everything else exchanges both planes at once.
template <typename T, bool PRE_TW, bool TRANSPOSE>
hipError_t launch_tile_mode(int lr, int lc, int lp, unsigned grid, hipStream_t stream, const TileArgs &a,
                            bool query_only, int *blocks_per_cu, size_t *lds, hipEvent_t e0, hipEvent_t e1) {
#define PHAST_CASE(LR_, LC_, LP_)                                                                                  \
    if (lr == LR_ && lc == LC_ && lp == LP_)                                                                       \
        return launch_tile_inst<T, LR_, LC_, LP_, PRE_TW, TRANSPOSE, plane_seq_v<T, LP_>>(                 \
            grid, stream, a, query_only, blocks_per_cu, lds, e0, e1);
    PHAST_TILE_SHAPES(PHAST_CASE)
    if constexpr (sizeof(T) == 4) {
        PHAST_TILE_SHAPES_F32(PHAST_CASE)
    }
#undef PHAST_CASE
    return hipErrorInvalidValue;
}

// the last pass of a real transform's inner FFT with the untangle fused in (r2c_fused.hpp): the same shapes, at most 16
// points per thread.  Included by tile_f64_r2c.hip / tile_f32_r2c.hip only (r2c_fused.hpp is not pulled in here).
#define PHAST_R2C_DISPATCH(T_)                                                                                              \
    template <> hipError_t launch_r2c_last<T_>(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a,      \
                                               const R2cFuseArgs &f, bool q, int *b, hipEvent_t e0, hipEvent_t e1) {          \
        PHAST_TILE_SHAPES(PHAST_R2C_CASE_##T_)                                                                              \
        return hipErrorInvalidValue;                                                                                        \
    }
struct R2cFuseArgs;
template <typename T>
hipError_t launch_r2c_last(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, const R2cFuseArgs &f, bool q, int *b,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// (two tiles' worth of points must fit the registers: f64 up to 16 points per thread on at most 512 threads, f32 up to 32
// points per thread on at most 512 threads; the rest keeps the sweep)
constexpr bool r2c_shape_fits(int lr, int lc, int lp, size_t elem_bytes) {
    return elem_bytes == 8 ? (lp <= 4 && lr + lc - lp <= 9) : (lp <= 4 || (lp == 5 && lr + lc - lp <= 9));
}
inline bool r2c_shape_ok(unsigned lr, unsigned lc, unsigned lp, size_t elem_bytes) {
    return r2c_shape_fits((int)lr, (int)lc, (int)lp, elem_bytes) && shape_exists(lr, lc, lp, elem_bytes);
}

// the first pass of an inverse real transform's inner FFT with the preprocess fused into its load (c2r_fused.hpp).
// Included by tile_f64_c2r.hip / tile_f32_c2r.hip only.
#define PHAST_C2R_DISPATCH(T_)                                                                                              \
    template <> hipError_t launch_c2r_first<T_>(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a,     \
                                                const C2rFuseArgs &f, bool q, int *b, hipEvent_t e0, hipEvent_t e1) {         \
        PHAST_TILE_SHAPES(PHAST_C2R_CASE_##T_)                                                                              \
        if constexpr (sizeof(T_) == 4) {                                                                                    \
            PHAST_TILE_SHAPES_F32(PHAST_C2R_CASE_##T_)                                                                      \
        }                                                                                                                   \
        return hipErrorInvalidValue;                                                                                        \
    }
struct C2rFuseArgs;
template <typename T>
hipError_t launch_c2r_first(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, const C2rFuseArgs &f, bool q, int *b,
                            hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// (a thread holds its own points and their partners while z is formed: shapes whose budget that exceeds keep the sweep)
constexpr bool c2r_shape_fits(int lr, int lc, int lp, size_t elem_bytes) {
    return !(elem_bytes == 8 && lr == 10 && lc == 4 && lp == 4);  // 1024 threads x 16 f64 points: 2 spilled registers
}
inline bool c2r_shape_ok(unsigned lr, unsigned lc, unsigned lp, size_t elem_bytes) {
    return c2r_shape_fits((int)lr, (int)lc, (int)lp, elem_bytes) && shape_exists(lr, lc, lp, elem_bytes);
}

// defined in tile_f64_a.hip, tile_f64_bc.hip, tile_f32_a.hip, tile_f32_bc.hip
hipError_t launch_tile_f64_a(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_tile_f64_bc(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_tile_f32_a(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_tile_f32_bc(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// defined in wave_f64.hip: one wave per 64 x 16 tile (wave_fft.hpp); transpose = first pass, else a pre-twiddle pass
hipError_t launch_wave_f64(bool transpose, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// defined in wave_f32.hip: the f32 wave tile, one wave per 64 x 32 tile, a lane holds float2 column pairs (round 6)
hipError_t launch_wave_f32(bool transpose, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// defined in quad_f64.hip / quad_f32.hip: four waves per 256 x 16 (f32: 256 x 32) tile, pre-twiddle passes (quad_fft.hpp)
hipError_t launch_quad_f64(unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_quad_f32(unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

}  // namespace phast
