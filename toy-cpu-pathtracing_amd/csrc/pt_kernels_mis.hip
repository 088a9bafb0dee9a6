// pt_kernel specialised for the MIS renderer with the ZSobol sampler (every BASELINE config but C1 and C5): the feature sets without the
// clearcoat code (pt_kernels_mis_cc.hip holds the others; the two translation units are compiled with different backend options, Makefile).
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_mis_sobol(const PtLaunchArgs& a, uint32_t feat) {
    if (pick_features(feat) & FEAT_CC) launch_pt_mis_sobol_cc(a, feat); else launch_pt_plain<MODE_MIS_SOBOL>(a, feat);
}
int occupancy_pt_mis_sobol(uint32_t feat) { return (pick_features(feat) & FEAT_CC) ? occupancy_pt_mis_sobol_cc(feat) : occupancy_pt_plain<MODE_MIS_SOBOL>(feat); }
}  // namespace pt
