// Instantiations of the fused head / operand kernel for DT in {12, 14, 16} (see prep_kernel.h).
#include "prep_kernel.h"

namespace pilco {

void launch_prep_12(const PrepLaunch& a) { launch_prep_dt<12>(a); }
void launch_prep_14(const PrepLaunch& a) { launch_prep_dt<14>(a); }
void launch_prep_16(const PrepLaunch& a) { launch_prep_dt<16>(a); }

}  // namespace pilco
