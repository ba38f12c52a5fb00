// The f32 tile-pass instantiation compiled with -fno-slp-vectorize -amdgpu-sched-strategy=max-ilp (tile_dispatch.hpp):
// the 1024 x 32 pre-twiddle pass at 32 points per thread on 1024 threads, which otherwise spills 19 VGPRs.
#include "tile_dispatch.hpp"
namespace phast {
template hipError_t launch_tile_inst<float, 10, 5, 5, true, false, true>(unsigned, hipStream_t, const TileArgs &, bool, int *, size_t *,
                                                                         hipEvent_t, hipEvent_t);
}  // namespace phast
