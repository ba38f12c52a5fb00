// wave_f32.hip -- the f32 wave-tile pass kernels (wave_fft.hpp: 64 rows x 32 columns, a lane holds float2 column pairs): first
// pass (transposing) and pre-twiddle passes.
#include "tile_dispatch.hpp"
#include "wave_fft.hpp"

namespace phast {
hipError_t launch_wave_f32(bool transpose, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l, hipEvent_t e0,
                           hipEvent_t e1) {
    return transpose ? launch_wave_inst<float, false, true>(s, a, q, b, l, e0, e1)
                     : launch_wave_inst<float, true, false>(s, a, q, b, l, e0, e1);
}
}  // namespace phast
