// Instantiations of the fused head / operand kernel for DT in {20, 24, 32} (see prep_kernel.h).
#include "prep_kernel.h"

namespace pilco {

void launch_prep_20(const PrepLaunch& a) { launch_prep_dt<20>(a); }
void launch_prep_24(const PrepLaunch& a) { launch_prep_dt<24>(a); }
void launch_prep_32(const PrepLaunch& a) { launch_prep_dt<32>(a); }

}  // namespace pilco
