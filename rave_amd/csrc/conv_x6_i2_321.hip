// conv_x6_kernel (conv_x6_kernel.inc), input stride 2, wave tile 96 x 64, 1 wave(s) along the rows: every
// (input activation, epilogue) instance of this shape.  One translation unit per shape so that the instances build in parallel.
#include <mutex>
#include <type_traits>
#include "conv_params.hpp"
#include "conv_x6_kernel.inc"

bool rh_x6_launch_i2_321(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    return x6_launch<2, 3, 2, 1>(q, epi, grid, lds, stream);
}
