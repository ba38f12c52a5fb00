// the f64 instantiations of the fused C2R first pass (c2r_fused.hpp)
#include "tile_dispatch.hpp"
#include "c2r_fused.hpp"
namespace phast {
#define PHAST_C2R_CASE_double(LR_, LC_, LP_)                                                                          \
    if constexpr (c2r_shape_fits(LR_, LC_, LP_, sizeof(double))) {                                                      \
        if (lr == LR_ && lc == LC_ && lp == LP_)                                                                     \
            return launch_c2r_first_inst<double, LR_, LC_, LP_, plane_seq_v<double, LP_>>(grid, s, a, f, q, b, e0, e1);    \
    }
PHAST_C2R_DISPATCH(double)
}  // namespace phast
