// pt_kernel specialised for the plain path tracer (strategy pt, either sampler: BASELINE configs[0] is pt + random): no light connections
// exist, so these kernels keep the closest-hit traversal alone (pt_kernel.hpp, merged_traversal) and fold the strategy branches away.
#include "pt_kernel.hpp"
namespace pt {
void launch_pt_strategy_pt(const PtLaunchArgs& a, uint32_t feat) { launch_pt_mode<MODE_PT>(a, feat); }
int occupancy_pt_strategy_pt(uint32_t feat) { return occupancy_pt_mode<MODE_PT>(feat); }
}  // namespace pt
