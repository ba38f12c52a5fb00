// the f32 instantiations of the fused R2C last pass (r2c_fused.hpp): every tile shape with at most 16 points per thread
#include "tile_dispatch.hpp"
#include "r2c_fused.hpp"
namespace phast {
#define PHAST_R2C_CASE_float(LR_, LC_, LP_)                                                                       \
    if constexpr (r2c_shape_fits(LR_, LC_, LP_, sizeof(float))) {                                                                                     \
        if (lr == LR_ && lc == LC_ && lp == LP_)                                                                  \
            return launch_r2c_last_inst<float, LR_, LC_, LP_, plane_seq_v<float, LP_>>(grid, s, a, f, q, b, e0, e1); \
    }
PHAST_R2C_DISPATCH(float)
}  // namespace phast
