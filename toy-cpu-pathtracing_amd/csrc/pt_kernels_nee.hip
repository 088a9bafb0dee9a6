// pt_kernel specialised for the NEE renderer with the ZSobol sampler: the feature sets without the clearcoat code (C5's kernel is in
// pt_kernels_nee_cc.hip; the two translation units are compiled with different backend options, Makefile).
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_nee_sobol(const PtLaunchArgs& a, uint32_t feat) {
    if (pick_features(feat) & FEAT_CC) launch_pt_nee_sobol_cc(a, feat); else launch_pt_plain<MODE_NEE_SOBOL>(a, feat);
}
int occupancy_pt_nee_sobol(uint32_t feat) { return (pick_features(feat) & FEAT_CC) ? occupancy_pt_nee_sobol_cc(feat) : occupancy_pt_plain<MODE_NEE_SOBOL>(feat); }
}  // namespace pt
