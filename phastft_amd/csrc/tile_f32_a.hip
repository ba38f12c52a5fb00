// first pass of a multi-pass f32 FFT: strided tile in, digit-reversing contiguous runs out
#include "tile_dispatch.hpp"
namespace phast {
hipError_t launch_tile_f32_a(int lr, int lc, int lp, unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                             hipEvent_t e0, hipEvent_t e1) {
    return launch_tile_mode<float, false, true>(lr, lc, lp, grid, s, a, q, b, l, e0, e1);
}
}  // namespace phast
