// quad_f32.hip -- the f32 four-wave 256-row pass kernel (quad_fft.hpp): 256 rows x 32 columns, a lane holds float2 column pairs.
#include "tile_dispatch.hpp"
#include "quad_fft.hpp"

namespace phast {
hipError_t launch_quad_f32(unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l, hipEvent_t e0,
                           hipEvent_t e1) {
    return launch_quad_inst<float>(grid, s, a, q, b, l, e0, e1);
}
}  // namespace phast
