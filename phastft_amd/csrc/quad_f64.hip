// quad_f64.hip -- the f64 four-wave 256-row pass kernel (quad_fft.hpp).
#include "tile_dispatch.hpp"
#include "quad_fft.hpp"

namespace phast {
hipError_t launch_quad_f64(unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l, hipEvent_t e0,
                           hipEvent_t e1) {
    return launch_quad_inst<double>(grid, s, a, q, b, l, e0, e1);
}
}  // namespace phast
