// bf16x6 convolution kernels for input stride 2 (see conv_x6_kernel.inc).
#include <mutex>
#include <type_traits>
#include "conv_params.hpp"
#include "conv_x6_kernel.inc"

void rh_x6_dispatch_is2(const ConvP& q, int tm, int tn, int wm, dim3 grid, size_t lds, hipStream_t stream) {
    x6_dispatch<2>(q, tm, tn, wm, grid, lds, stream);
}
