// pt_kernel specialised for the NEE renderer with the ZSobol sampler (C5): the feature sets with the clearcoat code (3 waves per SIMD).
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_nee_sobol_cc(const PtLaunchArgs& a, uint32_t feat) { launch_pt_cc<MODE_NEE_SOBOL>(a, feat); }
int occupancy_pt_nee_sobol_cc(uint32_t feat) { return occupancy_pt_cc<MODE_NEE_SOBOL>(feat); }
}  // namespace pt
