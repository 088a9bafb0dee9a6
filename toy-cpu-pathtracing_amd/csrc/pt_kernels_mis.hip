// pt_kernel specialised for the MIS renderer with the ZSobol sampler (every BASELINE config but C1 and C5), all feature sets.
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_mis_sobol(const PtLaunchArgs& a, uint32_t feat) { launch_pt_mode<MODE_MIS_SOBOL>(a, feat); }
}  // namespace pt
