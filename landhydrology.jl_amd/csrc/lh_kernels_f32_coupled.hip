// gfx950 column kernels (rhs_kernel), float, coupled model
#define LH_TU_MODEL
#include "lh_kernels_impl.hpp"
namespace lh {
LH_INSTANTIATE_MODEL(float, MODEL_COUPLED)
}
