// pt_kernel specialised for the NEE renderer with the ZSobol sampler (BASELINE C5), all feature sets.
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_nee_sobol(const PtLaunchArgs& a, uint32_t feat) { launch_pt_mode<MODE_NEE_SOBOL>(a, feat); }
}  // namespace pt
