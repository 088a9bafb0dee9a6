// pt_kernel specialised for the MIS renderer with the ZSobol sampler: the feature sets with the clearcoat code (3 waves per SIMD).
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_mis_sobol_cc(const PtLaunchArgs& a, uint32_t feat) { launch_pt_cc<MODE_MIS_SOBOL>(a, feat); }
int occupancy_pt_mis_sobol_cc(uint32_t feat) { return occupancy_pt_cc<MODE_MIS_SOBOL>(feat); }
}  // namespace pt
