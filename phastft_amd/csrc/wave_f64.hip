// wave_f64.hip -- the f64 wave-tile pass kernels (wave_fft.hpp): first pass (transposing) and pre-twiddle passes.
#include "tile_dispatch.hpp"
#include "wave_fft.hpp"

namespace phast {
hipError_t launch_wave_f64(bool transpose, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l, hipEvent_t e0,
                           hipEvent_t e1) {
    return transpose ? launch_wave_inst<double, false, true>(s, a, q, b, l, e0, e1)
                     : launch_wave_inst<double, true, false>(s, a, q, b, l, e0, e1);
}
}  // namespace phast
