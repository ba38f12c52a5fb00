// The one tile-pass instantiation that is compiled with -amdgpu-sched-strategy=max-ilp (see tile_dispatch.hpp): the
// 1024 x 16 f64 pre-twiddle pass at 32 points per thread, which otherwise spills registers to scratch memory.
#include "tile_dispatch.hpp"
namespace phast {
template hipError_t launch_tile_inst<double, 10, 4, 5, true, false, true>(unsigned, hipStream_t, const TileArgs &, bool, int *, size_t *,
                                                                          hipEvent_t, hipEvent_t);
}  // namespace phast
