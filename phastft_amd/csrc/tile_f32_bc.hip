// later passes of a multi-pass f32 FFT: inter-pass twiddle on load, same strided pattern in and out
#include "tile_dispatch.hpp"
namespace phast {
hipError_t launch_tile_f32_bc(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                             hipEvent_t e0, hipEvent_t e1) {
    return launch_tile_mode<float, true, false>(lr, lc, lp, grid, s, a, q, b, l, e0, e1);
}
}  // namespace phast
