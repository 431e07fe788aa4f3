// gfx950 kernels, Float64 instantiation
#include "lh_kernels_impl.hpp"
namespace lh {
LH_INSTANTIATE(double)
}
