// gfx950 kernels, Float32 instantiation
#include "lh_kernels_impl.hpp"
namespace lh {
LH_INSTANTIATE(float)
}
