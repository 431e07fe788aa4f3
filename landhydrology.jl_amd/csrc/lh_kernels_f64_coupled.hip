// gfx950 column kernels (rhs_kernel), double, coupled model
#define LH_TU_MODEL
#include "lh_kernels_impl.hpp"
namespace lh {
LH_INSTANTIATE_MODEL(double, MODEL_COUPLED)
}
